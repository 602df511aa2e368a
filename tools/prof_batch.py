#!/usr/bin/env python3
"""A pipelined batch of evaluations for rocprofv3 --kernel-trace:  prof_batch.py n n_evals slots"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

n, ne, slots = (int(a) for a in (sys.argv[1:] + ["8192", "8", "4"])[:3])
ctx = gsum_amd.lab_context(0)
desc = gsum_amd.describe_kernel(RBF(0.2), 1)
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
ctx.set_inputs(X, Z)
ctx.set_option("batch_slots", slots)
ctx.lml_resident([desc] * slots, 1e-10)
t0 = time.perf_counter()
ctx.lml_resident([desc] * ne, 1e-10)
print("ms/eval", (time.perf_counter() - t0) / ne * 1e3, flush=True)
