#!/usr/bin/env python3
"""Isolated launches of the kernel-matrix build for rocprofv3 --pmc:  prof_build.py n reps"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

n, reps = (int(a) for a in (sys.argv[1:] + ["8192", "3"])[:2])
ctx = gsum_amd.lab_context(0)
X = 0.1 * np.arange(n)[:, None]
desc = gsum_amd.describe_kernel(RBF(0.2), 1)
for _ in range(reps):
    m = ctx.kernel_matrix_dev(desc, X, diag_add=1e-10)
    m.free()
print("built", n, reps, flush=True)
