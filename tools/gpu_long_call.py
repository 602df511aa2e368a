#!/usr/bin/env python3
"""One long call of the batch entry point (gsum_lml_resident: many rounds of the groups) at n = 8192: evaluations per second for the
phase shift between groups (wave_shift), own-stream tails (wave_tail_rows) and in-phase rounds; bit-identity against the default.
Usage: gpu_long_call.py [n] [evaluations]        -> profiles/r05_long_call.log"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 192
ctx = gsum_amd.lab_context(0)
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
ctx.set_inputs(X, Z)
descs = ctx.desc_array([gsum_amd.describe_kernel(RBF(float(e)), 1) for e in np.linspace(0.15, 0.25, N)])
ctx.set_option("wave_long_rounds", 1 << 20)          # nothing automatic: the settings below are explicit
ctx.set_option("wave_shift", 0)
ctx.set_option("wave_tail_rows", 0)
ctx.set_option("wave_cohorts", 1)
ref = ctx.lml_resident(descs, 1e-10)
bad = 0
SETTINGS = ((0, 0), (0, -2048), (0, -3072), (2, 0), (2, -2048), (2, -3072), (3, -3072), (4, -3072), (3, -4096), (4, -4096), (2, -4096), (3, -2048), (0, 0))
if len(sys.argv) > 3:
    SETTINGS = tuple(tuple(int(v) for v in a.split(',')) for a in sys.argv[3:])
for setting in SETTINGS:
    shift, tail = setting[:2]
    coh = setting[2] if len(setting) > 2 else 1
    ctx.set_option("wave_shift", shift)
    ctx.set_option("wave_tail_rows", tail)
    ctx.set_option("wave_cohorts", coh)
    if len(setting) > 3:
        ctx.set_option("wave_size", setting[3])
    ts = []
    for _ in range(3 if N <= 400 else 2):
        t0 = time.perf_counter()
        got = ctx.lml_resident(descs, 1e-10)
        ts.append(time.perf_counter() - t0)
    same = all(np.array_equal(a, b) for a, b in zip(got, ref))
    bad += not same
    print(f"n={n} {N} evaluations per call  shift={shift} tail_rows={-tail} cohorts={coh} size={ctx.get_option('wave_size')}: {N / min(ts):7.1f} evals/s (best of 3; median {N / np.median(ts):7.1f})  "
          f"identical={same}", flush=True)
sys.exit(1 if bad else 0)
