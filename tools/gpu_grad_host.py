#!/usr/bin/env python3
"""Host share of ConjugateGaussianProcess.log_marginal_likelihood_batch (value + gradient of several thetas, one pipelined device
batch): the call as shipped, and beside it the per-theta scikit-learn clone + describe it used to do, timed in the same process."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from gsum_amd.kernels import describe_gradient, describe_kernel  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, ConstantKernel as C, Matern, WhiteKernel  # noqa: E402

for n, kern, d in ((1024, C(1.0) * RBF(0.2), 1), (2048, C(1.0) * RBF(0.2), 1), (2048, C(2.0) * Matern([0.7, 1.3], nu=2.5) + WhiteKernel(1e-6), 2)):
    rng = np.random.RandomState(n)
    X = 0.1 * np.arange(n)[:, None] if d == 1 else rng.rand(n, 2) * np.array([0.35, 0.65]) * np.sqrt(n)
    y = rng.randn(n, 4)
    gp = gsum_amd.ConjugateGaussianProcess(kernel=kern, center=0, disp=0, df=1, scale=1, optimizer=None, nugget=1e-8)
    gp.fit(X, y)
    thetas = [gp.kernel_.theta + 0.01 * i for i in range(8)]
    single = [gp.log_marginal_likelihood(t, eval_gradient=True) for t in thetas]
    batch = gp.log_marginal_likelihood_batch(thetas)
    same = all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(single, batch))
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        gp.log_marginal_likelihood_batch(thetas)
        ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for _ in range(10):
        ks = [gp.kernel_.clone_with_theta(t) for t in thetas]
        [describe_gradient(k, d) for k in ks]
        [describe_kernel(k, d) for k in ks]
    old_host = (time.perf_counter() - t0) / 10
    print(f"n={n} {type(kern).__name__}/{len(thetas[0])} params: batch of 8 {min(ts) * 1e3:.2f} ms best, {np.median(ts) * 1e3:.2f} median; "
          f"the removed clone+describe step alone {old_host * 1e3:.2f} ms per call; identical to single calls {same}", flush=True)
