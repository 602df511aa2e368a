"""Synthetic partial-sum data on MI355X — host-side mirror of gsum/datasets.py.

The reference draws the coefficient curves with ``scipy.stats.multivariate_normal.rvs`` (datasets.py:69-70), an
eigendecomposition of the n x n covariance on the CPU: minutes at n = 8192.  Here the covariance is built and
Cholesky-factorised on the device (the likelihood path's own kernels) and the draws are ``mean + L z``
(``gsum_tri_multiply``).  The draws are statistically the same process; they are not the reference's numbers for
the same seed (a different square root of K, a different normal stream), and the factorisation needs a positive
definite K: pass ``nugget > 0`` where the reference's ``allow_singular=True`` would have tolerated a singular one.
"""
from __future__ import annotations

import numpy as np
from sklearn.gaussian_process.kernels import RBF
from sklearn.utils import check_random_state

from ._lib import default_context
from .kernels import describe_kernel
from .series import partials

__all__ = ["make_gaussian_partial_sums", "make_gaussian_partial_sums_uniform", "make_gaussian_partial_sums_on_grid",
           "sample_mvn_cholesky"]


def sample_mvn_cholesky(kernel, X, n_draws, mean=None, nugget=0.0, random_state=0, device=None):
    """(n, n_draws) draws from N(mean, kernel(X) + nugget I) through the device Cholesky."""
    X = np.asarray(X, dtype=float)
    n = X.shape[0]
    ctx = default_context(device)
    L = ctx.kernel_matrix_dev(describe_kernel(kernel, X.shape[1]), X, diag_add=float(nugget))
    try:
        info = ctx.potrf(L)
        if info != 0:
            raise np.linalg.LinAlgError(
                "covariance is not positive definite to working precision (leading minor %d): the device sampler "
                "factorises with Cholesky, pass nugget > 0" % info)
        z = check_random_state(random_state).standard_normal((n, int(n_draws)))
        draws = ctx.tri_multiply(L, z)
    finally:
        L.free()
    if mean is not None:
        draws = draws + np.asarray(mean, dtype=float)[:, None]
    return draws


def make_gaussian_partial_sums(X, orders=5, kernel=None, mean=None, ratio=0.3, ref=1., nugget=0, random_state=0,
                               allow_singular=True, device=None):
    """Partial sums of Gaussian-process coefficient curves at X; same arguments as datasets.py:8-72
    (``allow_singular`` is accepted and ignored: see the module docstring)."""
    if kernel is None:
        kernel = RBF(0.5)                                        # datasets.py:52-53
    if mean is None:
        def mean(a):
            return np.zeros(a.shape[0])                          # datasets.py:54-56
    if isinstance(orders, int):
        orders = np.arange(orders)                               # datasets.py:58-59
    if callable(ratio):
        ratio = ratio(X)
    if callable(ref):
        ref = ref(X)
    coeffs = sample_mvn_cholesky(kernel, X, len(orders), mean=mean(X), nugget=nugget, random_state=random_state,
                                 device=device)
    return partials(coeffs=coeffs, ratio=ratio, ref=ref, orders=orders)          # datasets.py:71


def make_gaussian_partial_sums_uniform(n_samples=100, n_features=1, orders=5, kernel=None, mean=None, ratio=0.3, ref=1.,
                                       nugget=0, random_state=0, allow_singular=True, device=None):
    """Inputs drawn uniformly from [0, 1]^n_features (datasets.py:75-128)."""
    generator = check_random_state(random_state)
    X = generator.rand(n_samples, n_features)
    y = make_gaussian_partial_sums(X=X, orders=orders, kernel=kernel, mean=mean, ratio=ratio, ref=ref, nugget=nugget,
                                   random_state=random_state, allow_singular=allow_singular, device=device)
    return X, y


def make_gaussian_partial_sums_on_grid(n_samples=100, n_features=1, orders=5, kernel=None, mean=None, ratio=0.3, ref=1.,
                                       nugget=0, random_state=0, allow_singular=True, device=None):
    """Inputs on a full grid of n_samples points per feature in [0, 1] (datasets.py:131-190).  The reference builds
    the n_features > 1 grid from ``range(n_features)`` instead of the linspace; the intended
    Cartesian product of the linspace is used here."""
    x = np.linspace(0, 1, n_samples)
    if n_features > 1:
        X = np.stack(np.meshgrid(*[x] * n_features, indexing='ij'), -1).reshape(-1, n_features)   # helpers.py:19-33
    else:
        X = x[:, None]
    y = make_gaussian_partial_sums(X=X, orders=orders, kernel=kernel, mean=mean, ratio=ratio, ref=ref, nugget=nugget,
                                   random_state=random_state, allow_singular=allow_singular, device=device)
    return X, y
