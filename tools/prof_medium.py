#!/usr/bin/env python3
"""BASELINE configs[1] on the one-workgroup-per-evaluation path under a profiler: n = 2048, 4 orders, 512 evaluations per launch of
k_lml_medium (one round of the 512 resident workgroups).   rocprofv3 --kernel-trace [--pmc ...] -- python3 tools/prof_medium.py [n] [evals] [calls]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 2
r = 4
X = 0.1 * np.arange(n)[:, None]
c = np.random.RandomState(0).randn(n, r)
Z = np.concatenate([c, np.ones((n, 1))], axis=1)
ctx = gsum_amd.default_context(0)
ctx.set_inputs(X, Z)
descs = ctx.desc_array([gsum_amd.describe_kernel(RBF(float(e)), 1) for e in np.linspace(0.15, 0.25, K)])
ctx.lml_resident(descs, 1e-10)
for _ in range(calls):
    t0 = time.perf_counter()
    _, _, info = ctx.lml_resident(descs, 1e-10)
    dt = time.perf_counter() - t0
    print(f"n={n} {K} evaluations: {dt * 1e3:.2f} ms, {K / dt:.0f} evals/s, {K * n ** 3 / 3 / dt / 1e12:.1f} TF/s of Cholesky flops, failed {int(np.count_nonzero(info))}", flush=True)
