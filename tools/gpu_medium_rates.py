#!/usr/bin/env python3
"""Throughput of the fused one-workgroup-per-evaluation paths (n <= 128 and 128 < n <= 4096) and of value + gradient."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from gsum_amd.kernels import describe_gradient, describe_kernel  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C  # noqa: E402

ctx = gsum_amd.lab_context(0)
for n, cnt in ((100, 8192), (256, 1024), (512, 1024), (1024, 1024), (2048, 1024), (4096, 512)):
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, 4), np.ones((n, 1))], axis=1)
    ctx.set_inputs(X, Z)
    # marshalled once (ctypes array): per call the Python-side packing of thousands of descriptors costs more than the
    # small-n kernel itself (0.4 us per descriptor against 0.3 us per evaluation)
    descs = ctx.desc_array([describe_kernel(RBF(0.2 + 1e-5 * i), 1) for i in range(cnt)])
    ctx.lml_resident(descs, 1e-10)
    t0 = time.perf_counter()
    G, sld, info = ctx.lml_resident(descs, 1e-10)
    dt = time.perf_counter() - t0
    print(f"n={n:5d} evals={cnt:5d}  {cnt / dt:10.0f} evals/s  {dt / cnt * 1e6:8.2f} us each  {n ** 3 / 3 * cnt / dt / 1e12:6.2f} TF/s  info0={int(info[0])}", flush=True)
ctx.set_option("release_scratch", 1)
for n in (4096, 8192):
    kern = C(1.0) * RBF(0.2) + WhiteKernel(1e-10, noise_level_bounds="fixed")
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
    desc, prm = describe_kernel(kern, 1), describe_gradient(kern, 1)
    for la in (1, 0):
        ctx.set_option("lookahead", la)
        ctx.lml_grad(desc, prm, X, Z, 1e-10)
        t0 = time.perf_counter()
        for _ in range(4):
            out = ctx.lml_grad(desc, prm, X, Z, 1e-10)
        tg = (time.perf_counter() - t0) / 4 * 1e3
        ctx.lml_batch([desc], X, Z, 1e-10)
        t0 = time.perf_counter()
        for _ in range(4):
            ctx.lml_batch([desc], X, Z, 1e-10)
        tv = (time.perf_counter() - t0) / 4 * 1e3
        print(f"n={n} lookahead={la}: value+gradient {tg:.2f} ms, value alone {tv:.2f} ms, trace {out[3]}", flush=True)
ctx.set_option("lookahead", 1)
