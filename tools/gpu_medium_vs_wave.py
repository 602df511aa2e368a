#!/usr/bin/env python3
"""Many evaluations of a medium order in one call: the one-workgroup-per-evaluation kernel (k_lml_medium) against the grouped schedule with
two cohorts per group (round 5).  Usage: gpu_medium_vs_wave.py  -> profiles/r05_medium_vs_wave.log"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.lab_context(0)
for n, N in ((1024, 2048), (2048, 1024), (3072, 768), (4096, 512)):
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, 4), np.ones((n, 1))], axis=1)
    ctx.set_inputs(X, Z)
    descs = ctx.desc_array([gsum_amd.describe_kernel(RBF(float(e)), 1) for e in np.linspace(0.15, 0.25, N)])
    ref = None
    for label, opts in (("medium kernel", dict(medium_path=1)), ("grouped, 1 cohort", dict(medium_path=0, wave_cohorts=1)),
                        ("grouped, 2 cohorts", dict(medium_path=0, wave_cohorts=2)),
                        ("grouped, 2 cohorts x 12", dict(medium_path=0, wave_cohorts=2, wave_size=12)),
                        ("grouped, 2 cohorts x 16", dict(medium_path=0, wave_cohorts=2, wave_size=16)),
                        ("grouped, 2 cohorts x 24", dict(medium_path=0, wave_cohorts=2, wave_size=24))):
        ctx.set_option("wave_size", 8)
        for k, v in opts.items():
            ctx.set_option(k, v)
        got = ctx.lml_resident(descs, 1e-10)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            got = ctx.lml_resident(descs, 1e-10)
            ts.append(time.perf_counter() - t0)
        same = ref is None or all(np.array_equal(a, b) for a, b in zip(got, ref))
        ref = ref or got
        print(f"n={n} {N} evaluations, {label:24s}: {N / min(ts):9.1f} evals/s  {N * n ** 3 / 3 / min(ts) / 1e12:5.1f} TF/s  identical={same}", flush=True)
    ctx.set_option("medium_path", 1)
    ctx.set_option("wave_size", 8)
    ctx.set_option("wave_cohorts", 2)
    ctx.set_option("release_scratch", 1)
