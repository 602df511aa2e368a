#!/usr/bin/env python3
"""BASELINE config 5's predictive sweep (n = 16384 2-D Matern + White, m = 2048 new points): milliseconds per gsum_predict_terms call for
the grouping depth of the trailing updates (predict_depth), one or two half-sweeps (predict_split), k_panel256 or three launches per pair."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsum_amd
from sklearn.gaussian_process.kernels import Matern, WhiteKernel
n, m = 16384, 2048
side = np.array([0.35, 0.65]) * np.sqrt(n)
X = np.random.RandomState(0).rand(n, 2) * side
Xs = np.random.RandomState(1).rand(m, 2) * side
rhs = np.concatenate([np.random.RandomState(2).randn(n, 8), np.ones((n, 1))], axis=1)
lab = gsum_amd.lab_context(0)
desc = gsum_amd.describe_kernel(Matern([0.7, 1.3], nu=2.5) + WhiteKernel(1e-6), 2)
L, info = lab.factorize(desc, X, diag_add=1e-10)
ref = None
for depth, split, p256, la in ((2, 0, 1, 0), (2, 0, 1, 1), (3, 0, 1, 1), (4, 0, 1, 1), (6, 0, 1, 1), (2, 0, 1, 0), (4, 0, 1, 1), (2, 0, 1, 1)):
    lab.set_option("predict_depth", depth); lab.set_option("predict_split", split); lab.set_option("predict_panel256", p256); lab.set_option("predict_lookahead", la)
    got = lab.predict_terms(L, desc, X, Xs, rhs=rhs)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); got = lab.predict_terms(L, desc, X, Xs, rhs=rhs); ts.append(time.perf_counter() - t0)
    same = ref is None or (np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]))
    ref = ref or got
    print(f"depth={depth} split={split} panel256={p256} lookahead={la}: {min(ts) * 1e3:.2f} ms (median {np.median(ts) * 1e3:.2f}), {n * n * m / min(ts) / 1e12:.1f} TF/s, identical={same}", flush=True)
L.free()
