#!/usr/bin/env python3
"""The notebook's 80 x 100 scan (n = 5) with a flattened kernel and with RBF + RBF + White (a tree): wall time of the grid call and of the device call alone."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gsum_amd  # noqa: E402
from conftest import load_golden  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, WhiteKernel  # noqa: E402

g = load_golden("notebook_grid.json")
X, y = np.array(g["X_train"]), np.array(g["y_train"])
flat = RBF(0.2) + WhiteKernel(g["nugget"], noise_level_bounds="fixed")
tree = RBF(0.2) + RBF(2.5, length_scale_bounds="fixed") + WhiteKernel(g["nugget"], noise_level_bounds="fixed")
for name, kern in (("flat", flat), ("tree", tree), ("flat", flat), ("tree", tree)):
    gp = gsum_amd.TruncationGP(kernel=kern, ref=g["ref"], ratio=0.5, center=0, disp=0, df=1, scale=1, optimizer=None)
    gp.fit(X, y, orders=np.array(g["orders"]))
    thetas = [[t] for t in np.log(g["ls_vals"])]
    gp.log_marginal_likelihood_grid(thetas, g["ratio_vals"], mode="full")
    best = np.inf
    for _ in range(5):
        t0 = time.perf_counter()
        gp.log_marginal_likelihood_grid(thetas, g["ratio_vals"], mode="full")
        best = min(best, time.perf_counter() - t0)
    ctx = gp.coeffs_process._context()
    t0 = time.perf_counter()
    descs = gsum_amd.describe_thetas(kern, thetas, 1)
    t_desc = time.perf_counter() - t0
    arr = ctx.desc_array(descs * 80)
    Z = np.concatenate([np.random.RandomState(0).randn(5, 4), np.ones((5, 1))], axis=1)
    ctx.set_inputs(X, Z)
    ctx.lml_resident(arr, 1e-10)
    bd = np.inf
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.lml_resident(arr, 1e-10)
        bd = min(bd, time.perf_counter() - t0)
    print(f"{name}: grid call {best * 1e3:.2f} ms; describe_thetas(100) {t_desc * 1e3:.2f} ms; device call of 8000 evaluations {bd * 1e3:.2f} ms", flush=True)
