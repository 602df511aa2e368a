#!/usr/bin/env python3
"""The predictive sweep V^T = K(Xs, X) L^-T with its trailing updates paired (option predict_lazy: K = 512 every other step) against one
K = 256 update per step: predict(return_std) of m = 2048 new points on n = 16384 2-D points (BASELINE config 5's shape) and on n = 8192 1-D
points, same process, interleaved; means and standard deviations must be bit-identical."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, Matern, WhiteKernel  # noqa: E402

ctx = gsum_amd.lab_context(0)
for n, d, m in ((16384, 2, 2048), (8192, 1, 2048), (8192, 1, 4096)):
    rng = np.random.RandomState(n + m)
    if d == 2:
        side = np.array([0.35, 0.65]) * np.sqrt(n)
        X, Xs = rng.rand(n, 2) * side, rng.rand(m, 2) * side
        kern = Matern(length_scale=[0.7, 1.3], nu=2.5) + WhiteKernel(1e-6, "fixed")
    else:
        X, Xs = 0.1 * np.arange(n)[:, None], 0.1 * (rng.rand(m, 1) * n)
        kern = RBF(0.2)
    y = rng.randn(n, 4)
    gp = gsum_amd.ConjugateGaussianProcess(kernel=kern, center=0, disp=0, df=1, scale=1, optimizer=None, nugget=1e-8)
    gp.fit(X, y)
    ref, out = None, []
    for rnd in range(3):
        for lazy in (0, 1):
            ctx.set_option("predict_lazy", lazy)
            gp.predict(Xs, return_std=True)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                mean, std = gp.predict(Xs, return_std=True)
                ts.append((time.perf_counter() - t0) * 1e3)
            key = (mean.tobytes(), std.tobytes())
            ref = ref or key
            out.append(f"{'paired' if lazy else 'plain '} {min(ts):6.2f} ms{'' if key == ref else ' DIFFERENT'}")
    print(f"n={n} d={d} m={m}: " + "  ".join(out), flush=True)
ctx.set_option("predict_lazy", 1)
