#!/usr/bin/env python3
"""Value + gradient evaluations at the reference's own sizes for rocprofv3 --kernel-trace --stats:  prof_grad_small.py [reps]
(k_grad_small: one launch per call; its average duration over n = 8, 20, 64, 128 in equal parts, and k_lml_small beside it)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from gsum_amd.kernels import describe_gradient, describe_kernel  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = gsum_amd.default_context(0)
kern = C(1.0) * RBF(0.5) + WhiteKernel(1e-6, noise_level_bounds="fixed")
for n in (8, 20, 64, 128):
    X = np.linspace(0, 1, n)[:, None] * (0.1 * n + 1.0)
    Z = np.concatenate([np.random.RandomState(0).randn(n, 4), np.ones((n, 1))], axis=1)
    desc, prm = describe_kernel(kern, 1), describe_gradient(kern, 1)
    for _ in range(reps):
        ctx.lml_grad(desc, prm, X, Z, 1e-10)
        ctx.lml_batch([desc], X, Z, 1e-10)
    print("n", n, "done", flush=True)
