#!/usr/bin/env python3
"""Where a small-n objective evaluation spends its host time: log_marginal_likelihood(theta, eval_gradient=True) and fit() with the default
optimiser at the reference's own sizes (n = 20 ... 500)."""
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

for n in (8, 20, 48, 64, 96, 128, 500, 2048):
    rng = np.random.RandomState(n)
    X = np.linspace(0, 1, n)[:, None] * (0.1 * n)
    kern = C(1.0) * RBF(0.5) + WhiteKernel(1e-6, noise_level_bounds="fixed")
    y = gsum_amd.sample_mvn_cholesky(kern, X, 4, nugget=1e-8, random_state=1)
    gp = gsum_amd.ConjugateGaussianProcess(kernel=kern, center=0, disp=0, df=1, scale=1, optimizer=None)
    gp.fit(X, y)
    theta = kern.theta + 0.1
    gp.log_marginal_likelihood(theta, eval_gradient=True)
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        gp.log_marginal_likelihood(theta, eval_gradient=True)
    t_grad = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        gp.log_marginal_likelihood(theta)
    t_val = (time.perf_counter() - t0) / reps
    gpo = gsum_amd.ConjugateGaussianProcess(kernel=kern, center=0, disp=0, df=1, scale=1)
    t0 = time.perf_counter()
    gpo.fit(X, y)
    t_fit = time.perf_counter() - t0
    print(f"n={n}: value {t_val * 1e6:.0f} us, value+gradient {t_grad * 1e6:.0f} us per call; fit with L-BFGS {t_fit * 1e3:.1f} ms", flush=True)
    if n in (20, 500):
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(reps):
            gp.log_marginal_likelihood(theta, eval_gradient=True)
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("cumulative")
        st.print_stats(14)
