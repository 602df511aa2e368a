#!/usr/bin/env python3
"""The full-size property of tests/test_gpu_parity.py::test_maximum_size_property_n24576 at larger orders: right-hand sides taken from K itself,
so G = Z^T K^-1 Z must return K[cols][:, cols] (build + factorisation + solve, no host copy of the matrix).   gpu_large_order.py n [n ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import Matern, WhiteKernel  # noqa: E402

ctx = gsum_amd.default_context(0)
for n in [int(a) for a in sys.argv[1:]] or [32768]:
    rng = np.random.RandomState(8)
    X = rng.rand(n, 2) * 50.0 * np.sqrt(n / 24576.0)
    kern = Matern(length_scale=1.0, nu=2.5) + WhiteKernel(1e-2, noise_level_bounds="fixed")
    cols = np.array([0, 1, 127, 128, n // 2 - 1, n - 1])
    Z = kern(X, X[cols])
    Z[cols, np.arange(len(cols))] += 1e-2
    desc = gsum_amd.describe_kernel(kern, 2)
    t0 = time.perf_counter()
    G, sld, info = ctx.lml_batch([desc], X, Z, 0.0)
    t1 = time.perf_counter()
    G2, sld2, info2 = ctx.lml_batch([desc], X, Z, 0.0)
    t2 = time.perf_counter()
    err = float(np.abs(G[0] - Z[cols]).max())
    print(f"n={n}: info {int(info[0])}, max |G - K[cols, cols]| = {err:.2e}, sum log diag {sld[0]:.6f}, first call {1e3 * (t1 - t0):.0f} ms, second {1e3 * (t2 - t1):.0f} ms "
          f"({n ** 3 / 3 / (t2 - t1) * 1e-12:.1f} TF/s of Cholesky flops, uploads included), repeat identical {bool(np.array_equal(G, G2) and sld[0] == sld2[0])}", flush=True)
    ctx.set_option("release_scratch", 1)
