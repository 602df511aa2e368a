#!/usr/bin/env python3
"""A call of exactly two evaluations must neither be slow nor slow down the batches that follow it in the process (it used to give a second slot its own
look-ahead streams: 24 live streams, beyond what the HIP runtime runs side by side).  Pipelined path only (medium_path = 0)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.lab_context(0)
ctx.set_option("batch_slots", 20)
ctx.set_option("medium_path", 0)
for n in (2048, 4096, 8192):
    big = 64 if n <= 4096 else 20
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, 4), np.ones((n, 1))], axis=1)
    ctx.set_inputs(X, Z)
    descs = [gsum_amd.describe_kernel(RBF(0.2 + 0.001 * i), 1) for i in range(big)]

    def run(c):
        ctx.lml_resident(descs[:c], 1e-10)
        t0 = time.perf_counter()
        r = ctx.lml_resident(descs[:c], 1e-10)
        return (time.perf_counter() - t0) * 1e3, r

    t_big0, ref = run(big)
    t1, r1 = run(1)
    t2, r2 = run(2)
    t3, r3 = run(3)
    t_big1, again = run(big)
    same = all(np.array_equal(a, b) for a, b in zip(ref, again)) and np.array_equal(r2[0], ref[0][:2]) and np.array_equal(r1[0], ref[0][:1]) and np.array_equal(r3[0], ref[0][:3])
    print(f"n={n}: {big} evaluations {t_big0:.1f} ms; then 1: {t1:.2f} ms, 2: {t2:.2f} ms, 3: {t3:.2f} ms; {big} again {t_big1:.1f} ms; identical {same}", flush=True)
