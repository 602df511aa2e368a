#!/usr/bin/env python3
"""Rates of the bulk tile as built (steady state, 200 back-to-back launches at K = 256 / 512; 3 back-to-back), the pipelined batch
(20 in flight, 40 evaluations per timing) and one factorisation alone, with a fingerprint of the results for comparison across builds."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.lab_context(0)
M = 7936
ctx.bench_gemm_nt(7, M, M, 256, tri=True, lda=8208, reps=50)
n = 8192
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
desc = gsum_amd.describe_kernel(RBF(0.2), 1)
ctx.set_inputs(X, Z)
for rnd in range(3):
    r = {}
    for K in (256, 512):
        r[f"K{K}x200"] = round(ctx.bench_gemm_nt(7, M, M, K, tri=True, lda=8208, reps=200)[0], 2)
        r[f"K{K}x3"] = round(float(np.median([ctx.bench_gemm_nt(7, M, M, K, tri=True, lda=8208, reps=3)[0] for _ in range(5)])), 2)
    r["M4096_K256x200"] = round(ctx.bench_gemm_nt(7, 4096, 4096, 256, tri=True, lda=8208, reps=200)[0], 2)
    ctx.set_option("batch_slots", 20)
    ctx.lml_resident([desc] * 20, 1e-10)
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        G, sld, info = ctx.lml_resident([desc] * 40, 1e-10)
        rates.append(40 / (time.perf_counter() - t0))
    r["batch20_evals_per_s"] = round(float(np.median(rates)), 1)
    ctx.set_option("batch_slots", 1)
    ts = []
    for _ in range(5):
        ctx.lml_resident([desc], 1e-10)
        ts.append(ctx.timers()["potrf_ms"])
    r["single_potrf_ms"] = round(min(ts), 3)
    r["fingerprint"] = (float(sld[0]).hex(), float(G[0, 0, 0]).hex(), float(G[-1, 2, 3]).hex())
    print(r, flush=True)
