#!/usr/bin/env python3
"""Throughput of the one-workgroup-per-evaluation path (k_lml_medium, 128 < n <= 4096) at the C ABI: evaluations per second and the
Cholesky flop rate, descriptors marshalled once, with a fingerprint of the results for comparison across builds."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.lab_context(0)
LAZY = [int(a) for a in sys.argv[1:]] or [64]              # e.g. `gpu_medium_rate.py 1 2 1 2`: option medium_lazy (grouping depth) alternated (same-process A/B)
for n, count in [(n, c) for n, c in ((512, 2048), (1024, 2048), (1536, 1024), (2048, 1024), (3072, 512), (4096, 512)) for _ in LAZY]:
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, 4), np.ones((n, 1))], axis=1)
    ctx.set_inputs(X, Z)
    lazy = LAZY.pop(0); LAZY.append(lazy)
    ctx.set_option("medium_lazy", lazy)
    darr = ctx.desc_array([gsum_amd.describe_kernel(RBF(float(e)), 1) for e in np.linspace(0.15, 0.25, count)])
    ctx.lml_resident(darr, 1e-10)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        G, sld, info = ctx.lml_resident(darr, 1e-10)
        ts.append(time.perf_counter() - t0)
    dt = min(ts)
    print(f"n={n:5d} medium_lazy={lazy}: {count / dt:9.1f} evals/s  {n ** 3 / 3.0 * count / dt / 1e12:6.2f} TF/s ({n ** 3 / 3.0 * count / dt / 1e12 / 78.6:5.3f} of 78.6)  "
          f"failed {int(np.count_nonzero(info))}  fingerprint {float(sld[7]).hex()} {float(G[-1, 1, 2]).hex()}", flush=True)
ctx.set_option("medium_lazy", 64)
