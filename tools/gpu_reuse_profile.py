#!/usr/bin/env python3
"""Host profile of BASELINE config 4's (cbar, ratio) grid in mode='reuse' (one factorisation, 64 x 64 points of host algebra)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
r = 6
X = 0.1 * np.arange(n)[:, None]
c = np.random.RandomState(0).randn(n, r)
y = gsum_amd.partials(c, ratio=0.5, ref=1.0, orders=np.arange(r))
gp = gsum_amd.TruncationGP(kernel=RBF(0.2), ratio=0.5, ref=1.0, center=0, disp=0, df=1, scale=1, optimizer=None)
gp.fit(X, y, orders=np.arange(r))
ratios = list(np.linspace(0.3, 0.7, 64))
cbars = np.geomspace(0.25, 4, 64)
theta = np.log([0.2])
for _ in range(2):
    gp.log_marginal_likelihood_grid([theta], ratios, scales=cbars, mode="reuse")
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    gp.log_marginal_likelihood_grid([theta], ratios, scales=cbars, mode="reuse")
    ts.append(time.perf_counter() - t0)
print("n=%d reuse grid 64 x 64: best %.2f ms" % (n, min(ts) * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    gp.log_marginal_likelihood_grid([theta], ratios, scales=cbars, mode="reuse")
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative")
st.print_stats(16)
