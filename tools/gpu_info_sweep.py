#!/usr/bin/env python3
"""Success / failure pattern of the device Cholesky against numpy.linalg.cholesky on near-singular inputs beyond the S0 class:
Matern-5/2 in one dimension (nugget 0 and 1e-14), RBF and Matern on random 2-D points."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, Matern  # noqa: E402

ctx = gsum_amd.lab_context(0)
ctx.set_option("medium_path", 0)
guard = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx.set_option("pivot_guard_ulps", guard)
print("pivot guard:", guard, "ulps", flush=True)
# exactly singular: a duplicated point, no nugget (the reference's non-PD case: LAPACK says not positive definite)
Xd = np.linspace(0, 1, 300)[:, None].copy()
Xd[200] = Xd[199]
for kern in (RBF(0.2), Matern(length_scale=0.3, nu=2.5), RBF(0.05)):
    K = kern(Xd)
    try:
        np.linalg.cholesky(K)
        lap = 0
    except np.linalg.LinAlgError:
        lap = 1
    _, _, info = ctx.lml_batch([gsum_amd.describe_kernel(kern, 1)], Xd, np.ones((300, 1)), 0.0)
    print("duplicated point:", kern, "LAPACK fails", bool(lap), "device info", int(info[0]), flush=True)
cases = []
n = 512
X1 = np.linspace(0, 1, n)[:, None]
rng = np.random.RandomState(0)
X2 = rng.rand(n, 2)
for name, X, kern_of, ells, nug in (
        ("matern52 1-D nugget 0", X1, lambda e: Matern(length_scale=e, nu=2.5), np.geomspace(0.5, 200, 48), 0.0),
        ("matern52 1-D nugget 1e-14", X1, lambda e: Matern(length_scale=e, nu=2.5), np.geomspace(0.5, 200, 48), 1e-14),
        ("rbf 2-D nugget 1e-12", X2, lambda e: RBF(length_scale=e), np.geomspace(0.05, 2, 48), 1e-12),
        ("rbf 2-D aniso nugget 1e-11", X2, lambda e: RBF(length_scale=[e, 2 * e]), np.geomspace(0.05, 2, 48), 1e-11),
        ("matern52 2-D nugget 0", X2, lambda e: Matern(length_scale=e, nu=2.5), np.geomspace(0.3, 100, 48), 0.0)):
    d = X.shape[1]
    want = []
    for e in ells:
        K = kern_of(float(e))(X)
        K[np.diag_indices_from(K)] += nug
        try:
            np.linalg.cholesky(K)
            want.append(True)
        except np.linalg.LinAlgError:
            want.append(False)
    descs = [gsum_amd.describe_kernel(kern_of(float(e)), d) for e in ells]
    _, _, info = ctx.lml_batch(descs, X, np.ones((n, 1)), nug)
    got = [int(i) == 0 for i in info]
    diff = [(round(float(e), 4), g, w) for e, g, w in zip(ells, got, want) if g != w]
    print(f"{name}: LAPACK ok {sum(want)}/{len(want)}, device ok {sum(got)}/{len(got)}, disagreements {len(diff)}: {diff}", flush=True)
