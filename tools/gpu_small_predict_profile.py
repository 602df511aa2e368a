#!/usr/bin/env python3
"""predict() at the reference's own sizes (n = 5 ... 100 training points, a few hundred new points): wall time per call on both backends,
and the host profile of the hip one."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C  # noqa: E402

for n, m in ((10, 200), (100, 500)):
    X = np.linspace(0, 1, n)[:, None] * (0.1 * n + 1.0)
    Xs = np.linspace(0, 1, m)[:, None] * (0.1 * n + 1.0)
    kern = C(1.0) * RBF(0.5) + WhiteKernel(1e-6, noise_level_bounds="fixed")
    y = gsum_amd.sample_mvn_cholesky(kern, X, 4, nugget=1e-8, random_state=1)
    orders = np.arange(4)
    yp = gsum_amd.partials(y, ratio=0.5, ref=1.0, orders=orders)
    for backend in ("hip", "cpu"):
        gp = gsum_amd.ConjugateGaussianProcess(kernel=kern, center=0, disp=0, df=1, scale=1, optimizer=None, backend=backend).fit(X, y)
        tg = gsum_amd.TruncationGP(kernel=kern, ratio=0.5, ref=1.0, center=0, disp=0, df=1, scale=1, optimizer=None, backend=backend)
        tg.fit(X, yp, orders=orders)
        res = {}
        for name, f in (("cgp.predict(std)", lambda: gp.predict(Xs, return_std=True)), ("cgp.predict(cov)", lambda: gp.predict(Xs, return_cov=True)),
                        ("trunc.predict(order 3, std)", lambda: tg.predict(Xs, order=3, return_std=True)), ("cgp.fit", lambda: gp.fit(X, y))):
            f()
            t0 = time.perf_counter()
            for _ in range(50):
                f()
            res[name] = (time.perf_counter() - t0) / 50 * 1e6
        print(f"n={n} m={m} {backend}: " + ", ".join(f"{k} {v:.0f} us" for k, v in res.items()), flush=True)
        if backend == "hip" and n == 10:
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(100):
                gp.predict(Xs, return_std=True)
            pr.disable()
            st = pstats.Stats(pr)
            st.sort_stats("cumulative")
            st.print_stats(14)
