#!/usr/bin/env python3
"""Value + gradient evaluations for rocprofv3 --kernel-trace:  prof_grad.py n [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from gsum_amd.kernels import describe_gradient, describe_kernel  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = gsum_amd.lab_context(0)
kern = C(1.0) * RBF(0.2) + WhiteKernel(1e-10, noise_level_bounds="fixed")
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
desc, prm = describe_kernel(kern, 1), describe_gradient(kern, 1)
ctx.lml_grad(desc, prm, X, Z, 1e-10)
t0 = time.perf_counter()
for _ in range(reps):
    out = ctx.lml_grad(desc, prm, X, Z, 1e-10)
print(n, "value+grad ms %.2f" % ((time.perf_counter() - t0) / reps * 1e3), "trace", out[3], flush=True)
