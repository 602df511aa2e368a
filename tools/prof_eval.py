#!/usr/bin/env python3
"""One fused likelihood evaluation per size, for rocprofv3 --kernel-trace:  prof_eval.py n [n ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.default_context(0)
desc = gsum_amd.describe_kernel(RBF(0.2), 1)
reps = int(os.environ.get("PROF_REPS", "3"))
for n in [int(a) for a in sys.argv[1:]] or [8192]:
    r = 6
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, r), np.ones((n, 1))], axis=1)
    ctx.set_inputs(X, Z)
    for _ in range(reps):
        G, sld, info = ctx.lml_resident([desc], 1e-10)
    print(n, ctx.timers(), int(info[0]), flush=True)
