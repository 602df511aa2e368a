#!/usr/bin/env python3
"""Value + gradient: eight evaluations one after the other (gsum_lml_grad) against one pipelined batch (gsum_lml_grad_batch)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from gsum_amd.kernels import describe_kernel, describe_gradient  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C  # noqa: E402

ctx = gsum_amd.lab_context(0) if (len(sys.argv) < 2 or sys.argv[1] != "product") else gsum_amd.HipContext(0)
print("library:", "product" if len(sys.argv) > 1 and sys.argv[1] == "product" else "lab", flush=True)
for n in (1024, 2048, 4096, 8192):
    r = 6
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, r), np.ones((n, 1))], axis=1)
    base = C(1.0) * RBF(0.2) + WhiteKernel(1e-8)
    kernels = [base.clone_with_theta(base.theta + 0.01 * i) for i in range(8)]
    params = [describe_gradient(k, 1) for k in kernels]
    descs = [describe_kernel(k, 1) for k in kernels]
    ctx.lml_grad(descs[0], params[0], X, Z, 1e-10)
    ctx.lml_grad_batch(descs, params, X, Z, 1e-10)
    t0 = time.perf_counter()
    single = [ctx.lml_grad(d, p, X, Z, 1e-10) for d, p in zip(descs, params)]
    t1 = time.perf_counter()
    batch = ctx.lml_grad_batch(descs, params, X, Z, 1e-10)
    t2 = time.perf_counter()
    same = all(np.array_equal(batch[0][i], single[i][0]) and np.array_equal(batch[4][i], single[i][4]) for i in range(8))
    print(f"n={n}: 8 single calls {1e3 * (t1 - t0):.2f} ms ({1e3 * (t1 - t0) / 8:.2f} each), one batch of 8 {1e3 * (t2 - t1):.2f} ms "
          f"({1e3 * (t2 - t1) / 8:.2f} each), ratio {(t1 - t0) / (t2 - t1):.2f}, identical {same}", flush=True)
    ctx.set_option("release_scratch", 1)
