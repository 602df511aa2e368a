#!/usr/bin/env python3
"""One batch call (K evaluations, n = 8192) under a profiler:  rocprofv3 --kernel-trace -- python3 tools/prof_wave.py G B depth [K]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

G, B, depth = (int(v) for v in sys.argv[1:4])
K = int(sys.argv[4]) if len(sys.argv) > 4 else 20
n, r = 8192, 6
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, r), np.ones((n, 1))], axis=1)
ctx = gsum_amd.lab_context(0)
ctx.set_inputs(X, Z)
ctx.set_option("wave_groups", G)
ctx.set_option("wave_size", B)
ctx.set_option("wave_depth", depth)
descs = ctx.desc_array([gsum_amd.describe_kernel(RBF(float(e)), 1) for e in np.linspace(0.19, 0.21, K)])
ctx.lml_resident(descs, 1e-10)
import time
t0 = time.perf_counter()
ctx.lml_resident(descs, 1e-10)
print("call ms", (time.perf_counter() - t0) * 1e3)
