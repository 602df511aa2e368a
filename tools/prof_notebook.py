#!/usr/bin/env python3
"""The notebook's 80 x 100 scan (n = 5, 8000 evaluations, one call) for rocprofv3 --kernel-trace --memory-copy-trace --stats:  prof_notebook.py [reps]
(what the device call's 1.4 ms are made of: k_lml_small against the copies of 8000 descriptors in and 8000 Gram matrices out)"""
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

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = load_golden("notebook_grid.json")
X, y = np.array(g["X_train"]), np.array(g["y_train"])
kern = RBF(0.2) + WhiteKernel(g["nugget"], noise_level_bounds="fixed")
gp = gsum_amd.TruncationGP(kernel=kern, ref=g["ref"], ratio=0.5, center=0, disp=0, df=1, scale=1, optimizer=None)
gp.fit(X, y, orders=np.array(g["orders"]))
thetas = [[t] for t in np.log(g["ls_vals"])]
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    gp.log_marginal_likelihood_grid(thetas, g["ratio_vals"], mode="full")
    ts.append(time.perf_counter() - t0)
print("grid call: best %.2f ms, median %.2f ms" % (min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3))
