"""CPU oracle for the gsum GP hot path.  TEST INFRASTRUCTURE ONLY.

This module is a CPU restatement of the algorithm that buqeye/gsum runs for
kernel build -> jittered Cholesky -> multivariate-normal log-likelihood behind
``ConjugateGaussianProcess`` / ``TruncationGP``.  It issues the same
numpy / scipy / scikit-learn calls, in the same order, as the reference does
(each function cites the reference ``file:line`` it follows), so that it can
stand in for the reference on machines where the reference is absent (the GPU
box).

Parity status: PINNED.  ``tests/golden/*.json`` hold outputs of the reference
itself (imported read-only in the build container by
``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` checks every
function below against them, and against the one likelihood-grid known answer
the reference publishes (notebook MAP indices (36, 39)).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module, and there only as the checker /
reported baseline.  The product package ``gsum_amd`` never imports it.
"""
from __future__ import annotations

import numpy as np
from numpy.linalg import cholesky, solve
from scipy.linalg import cho_solve, inv

__all__ = [
    "coefficients", "partials", "geometric_sum",
    "posterior_center", "posterior_disp", "posterior_df", "posterior_scale_sq",
    "cov_factor", "cgp_lml", "trunc_lml", "cgp_fit", "cgp_predict", "lml_grid",
    "trunc_mean", "trunc_cov", "trunc_basis", "trunc_predict_trunc", "cgp_prior_predict",
    "cbar_ratio_grid", "cbar_ratio_grid_one_factor",
]


# --------------------------------------------------------------------------
# series helpers  (reference: gsum/helpers.py)
# --------------------------------------------------------------------------

def coefficients(y, ratio, ref=1, orders=None):
    """y_n -> c_n.  Follows gsum/helpers.py:71-101.

    c_0 = y_0, c_n = y_n - y_{n-1}, then divided by ref * ratio**order.
    """
    y = np.asarray(y)
    if y.ndim != 2:
        raise ValueError("y must be 2d")                       # helpers.py:86-87
    if orders is None:
        orders = np.arange(y.shape[-1])                        # helpers.py:88-89
    if len(orders) != y.shape[-1]:
        raise ValueError("partials and orders must have the same length")
    ref, ratio, orders = np.atleast_1d(ref, ratio, orders)     # helpers.py:93
    ref = ref[:, None]
    ratio = ratio[:, None]
    c = np.diff(y, axis=-1)                                    # helpers.py:98
    c = np.insert(c, 0, y[..., 0], axis=-1)                    # helpers.py:99
    return c / (ref * ratio ** orders)                         # helpers.py:100


def partials(coeffs, ratio, ref=1, orders=None):
    """c_n -> partial sums y_k.  Follows gsum/helpers.py:104-146."""
    coeffs = np.asarray(coeffs)
    if orders is None:
        orders = np.arange(coeffs.shape[-1])
    ratio = np.atleast_1d(ratio)
    if ratio.ndim == 1:
        ratio = ratio[:, None]
    ref = np.atleast_1d(ref)
    if ref.ndim == 1:
        ref = ref[:, None]
    return np.cumsum(ref * coeffs * ratio ** orders, axis=-1)  # helpers.py:145-146


def geometric_sum(x, start, end, excluded=None):
    """sum_{i=start}^{end} x**i minus excluded orders.  helpers.py:149-182."""
    if end < start:
        raise ValueError("end must be greater than or equal to start")
    s = (x ** start - x ** (end + 1)) / (1 - x)                # helpers.py:176
    if excluded is not None:
        for n in np.atleast_1d(excluded):
            if start <= n <= end:
                s -= x ** n                                    # helpers.py:179-181
    return s


# --------------------------------------------------------------------------
# conjugate updates  (reference: gsum/models.py:170-503)
# --------------------------------------------------------------------------

def _num_y(y):                                                 # models.py:601-607
    return y.shape[1] if y.ndim == 2 else 1


def _avg_y(y):                                                 # models.py:609-628
    if y.ndim == 1:
        return np.copy(y)
    if y.ndim == 2:
        return np.average(y, axis=1)
    raise ValueError("y must be two-dimensional, not shape={}".format(y.shape))


def _rsolve(L, b):
    """R^{-1} b from the lower Cholesky factor.  models.py:478-479."""
    return cho_solve((L, True), b)


def posterior_disp(y, L, basis, disp0):
    """V = (V0^-1 + ny B^T R^-1 B)^-1, zero stays zero.  models.py:258-270."""
    if np.all(disp0 == 0):
        return np.zeros_like(disp0)
    ny = _num_y(y)
    quad = basis.T @ _rsolve(L, basis)                         # models.py:269
    return inv(inv(disp0) + ny * quad)                         # models.py:270


def posterior_center(y, L, basis, center0, disp0):
    """eta = V (V0^-1 eta0 + ny B^T R^-1 ybar).  models.py:199-220."""
    if np.all(disp0 == 0):
        return np.copy(center0)                                # models.py:201-206
    ybar = _avg_y(y)
    ny = _num_y(y)
    invR_ybar = _rsolve(L, ybar)                               # models.py:217
    disp = posterior_disp(y, L, basis, disp0)                  # models.py:218
    factor = solve(disp0, center0) + ny * basis.T @ invR_ybar  # models.py:219
    return disp @ factor                                       # models.py:220


def posterior_df(y, df0):
    return df0 + y.size                                        # models.py:302


def posterior_scale_sq(y, L, basis, center0, disp0, df0, scale0):
    """tau^2 update with the Woodbury mean term.  models.py:419-448."""
    if df0 == np.inf:
        return scale0 ** 2                                     # models.py:419-422
    if y.ndim == 1:
        y = y[:, None]
    ybar = _avg_y(y)
    N = len(ybar)
    ny = _num_y(y)
    yc = y - ybar[:, None]                                     # models.py:430
    quad = np.trace(yc.T @ _rsolve(L, yc))                     # models.py:432-433
    ybar_c = ybar - basis @ center0                            # models.py:435
    disp = posterior_disp(y, L, basis, disp0)                  # models.py:436
    invR_basis = _rsolve(L, basis)                             # models.py:438
    invR_ybar_c = _rsolve(L, ybar_c)                           # models.py:439
    mat = np.eye(N) - ny * invR_basis @ disp @ basis.T         # models.py:441
    mat_v = ny * mat @ invR_ybar_c                             # models.py:442
    quad2 = ybar_c @ mat_v                                     # models.py:445
    df = posterior_df(y, df0)
    return (df0 * scale0 ** 2 + quad + quad2) / df             # models.py:448


def cov_factor(scale_sq, df):
    """sigma^2 = nu tau^2/(nu-2), or tau^2 when nu = inf.  models.py:500-503."""
    if df != np.inf:
        return df * scale_sq / (df - 2)
    return scale_sq


def _ones_basis(X):
    return np.ones((X.shape[0], 1))                            # models.py:150


def _priors(center=0, disp=0, df=1, scale=1, sd=None):
    """Constructor prior bookkeeping.  models.py:113-120."""
    center0 = np.atleast_1d(center)
    disp0 = np.atleast_2d(disp)
    if sd is not None:
        return center0, disp0, np.inf, sd
    return center0, disp0, df, scale


# --------------------------------------------------------------------------
# log marginal likelihood  (reference: gsum/models.py:912-1057, 1485-1507)
# --------------------------------------------------------------------------

def cgp_lml(kernel, theta, X, y, center=0, disp=0, df=1, scale=1, sd=None,
            nugget=1e-10, return_parts=False):
    """ConjugateGaussianProcess.log_marginal_likelihood, value path.

    Follows models.py:953-1039 with decomposition='cholesky'.  ``kernel`` is a
    scikit-learn kernel; ``theta`` its log-hyperparameters (or None to keep
    the kernel as is).  Returns -inf when the Cholesky fails (models.py:970-972).
    """
    center0, disp0, df0, scale0 = _priors(center, disp, df, scale, sd)
    if theta is not None:
        kernel = kernel.clone_with_theta(np.asarray(theta, dtype=float))   # :953
    R = kernel(X)                                              # models.py:960
    R[np.diag_indices_from(R)] += nugget                       # models.py:963
    try:
        L = cholesky(R)                                        # models.py:969
    except np.linalg.LinAlgError:
        return -np.inf                                         # models.py:970-972
    if y.ndim == 1:
        y = y[:, np.newaxis]                                   # models.py:980-981
    dfn = posterior_df(y, df0)                                 # models.py:986
    basis = _ones_basis(X)                                     # models.py:987
    center_n = posterior_center(y, L, basis, center0, disp0)   # models.py:1001
    scale2 = posterior_scale_sq(y, L, basis, center0, disp0, df0, scale0)  # :1002
    mean = basis @ center_n                                    # models.py:1007
    var = cov_factor(scale2, dfn)                              # models.py:1008
    Lk = np.sqrt(var) * L                                      # models.py:1014
    logdet_K = 2 * np.log(np.diag(Lk)).sum()                   # models.py:1015
    y_train = y - mean[:, None]                                # models.py:1026
    N = R.shape[0]
    alpha = cho_solve((Lk, True), y_train)                     # models.py:1032
    ll = -0.5 * np.einsum("ik,ik->k", y_train, alpha)          # models.py:1035
    ll -= 0.5 * logdet_K                                       # models.py:1037
    ll -= N / 2 * np.log(2 * np.pi)                            # models.py:1038
    out = ll.sum(-1)                                           # models.py:1039
    if return_parts:
        return out, dict(L=L, center=center_n, scale_sq=scale2, df=dfn, var=var)
    return out


def trunc_lml(kernel, theta, X, y, orders, ratio=0.5, ref=1, excluded=None,
              center=0, disp=0, df=1, scale=1, sd=None, nugget=1e-10):
    """TruncationGP.log_marginal_likelihood.  Follows models.py:1485-1507.

    ``ratio`` / ``ref`` are scalars or (n,) arrays (the reference's default
    lambdas broadcast a scalar to (n,), models.py:1310,1315).
    """
    orders = np.asarray(orders)
    n = X.shape[0]
    ref_v = ref * np.ones(n) if np.ndim(ref) == 0 else np.asarray(ref)
    ratio_v = ratio * np.ones(n) if np.ndim(ratio) == 0 else np.asarray(ratio)
    mask = ~np.isin(orders, excluded)                          # models.py:1495
    c = coefficients(y, ratio_v, ref_v, orders)[:, mask]       # models.py:1496
    c_ll = cgp_lml(kernel, theta, X, c, center, disp, df, scale, sd, nugget)  # :1497
    orders_in = orders[mask]                                   # models.py:1503
    n_in = len(orders_in)
    det = np.sum(n_in * np.log(np.abs(ref_v))
                 + np.sum(orders_in) * np.log(np.abs(ratio_v)))  # models.py:1505
    return c_ll - det                                          # models.py:1506


def lml_grid(kernel, thetas, ratios, X, y, orders, ref=1, excluded=None, **priors):
    """The notebook's nested scan: rows = ratio values, columns = thetas.

    docs/notebooks/correlated_EFT_publication.ipynb:1457-1459.
    """
    out = np.empty((len(ratios), len(thetas)))
    for i, q in enumerate(ratios):
        for j, th in enumerate(thetas):
            out[i, j] = trunc_lml(kernel, np.atleast_1d(th), X, y, orders,
                                  ratio=q, ref=ref, excluded=excluded, **priors)
    return out


def cbar_ratio_grid(kernel, theta, X, y, orders, ratios, cbars, ref=1, excluded=None, center=0, disp=0,
                    nugget=1e-10):
    """BASELINE config 4 spelled the way the reference would: out[a, b] = one full
    TruncationGP(sd=cbars[b], ...).log_marginal_likelihood(theta, ratio=ratios[a]) call (models.py:115-117 makes
    ``sd`` the prior df0 = inf, scale0 = sd; :419-422 then fixes scale^2 = sd^2)."""
    out = np.empty((len(ratios), len(cbars)))
    for a, q in enumerate(ratios):
        for b, cbar in enumerate(cbars):
            out[a, b] = trunc_lml(kernel, theta, X, y, orders, ratio=q, ref=ref, excluded=excluded,
                                  center=center, disp=disp, sd=cbar, nugget=nugget)
    return out


def cbar_ratio_grid_one_factor(kernel, theta, X, y, orders, ratios, cbars, ref=1, excluded=None, center=0,
                               disp=0, nugget=1e-10, return_scale=False):
    """The same surface from ONE Cholesky factorisation, for checking large grids in seconds instead of hours.

    Every grid point shares R = kernel_theta(X) + nugget I (models.py:958-969 depend on theta only).  With
    sd = cbar the likelihood terms of models.py:1007-1039 are, per curve k,
        -1/2 (c_k - B eta)^T R^-1 (c_k - B eta) / cbar^2 - 1/2 (N log cbar^2 + 2 sum log L_ii) - N/2 log 2 pi,
    with c = coefficients(y, ratio, ref, orders) (helpers.py:98-100) and eta from compute_center (:199-220);
    the truncation layer subtracts the Jacobian of models.py:1503-1506.  The triangular solves are redone for
    every ratio (no rescaling identity is assumed), only the factor is shared.  Returns -inf everywhere if the
    factorisation fails (models.py:970-972).  ``return_scale=True`` also returns, per entry, the sum of the magnitudes
    of the terms the value is the signed sum of (quadratic form, log-determinant, constants, Jacobian): near the
    maximum of the surface these cancel to a small number, and a comparison tolerance must be relative to them."""
    orders = np.asarray(orders)
    n = X.shape[0]
    center0, disp0 = np.atleast_1d(center), np.atleast_2d(disp)
    k = kernel.clone_with_theta(np.asarray(theta, dtype=float)) if theta is not None else kernel
    R = k(X)
    R[np.diag_indices_from(R)] += nugget
    out = np.empty((len(ratios), len(cbars)))
    mag = np.empty_like(out)
    try:
        L = cholesky(R)
    except np.linalg.LinAlgError:
        out[:] = -np.inf
        return (out, np.ones_like(out)) if return_scale else out
    sld = np.log(np.diag(L)).sum()
    basis = _ones_basis(X)
    ref_v = ref * np.ones(n) if np.ndim(ref) == 0 else np.asarray(ref)
    mask = ~np.isin(orders, excluded)
    orders_in = orders[mask]
    n_in = len(orders_in)
    # residuals of every ratio side by side, then ONE cho_solve call for all of them (scipy copies the 0.5 GB factor
    # on every call: 64 calls took 4 minutes at n = 8192, one call takes seconds); column for column the same solve
    resids, dets = [], []
    for a, q in enumerate(ratios):
        ratio_v = q * np.ones(n)
        c = coefficients(y, ratio_v, ref_v, orders)[:, mask]
        eta = posterior_center(c, L, basis, center0, disp0)
        resids.append(c - (basis @ eta)[:, None])
        dets.append(np.sum(n_in * np.log(np.abs(ref_v)) + np.sum(orders_in) * np.log(np.abs(ratio_v))))
    R_all = np.concatenate(resids, axis=1)
    S_all = _rsolve(L, R_all)
    for a, q in enumerate(ratios):
        cols = slice(a * n_in, (a + 1) * n_in)
        quad = np.einsum("ik,ik->", R_all[:, cols], S_all[:, cols])
        det = dets[a]
        for b, cbar in enumerate(cbars):
            var = cbar ** 2
            out[a, b] = (-0.5 * quad / var - 0.5 * n_in * (n * np.log(var) + 2.0 * sld)
                         - n_in * n / 2 * np.log(2 * np.pi)) - det
            mag[a, b] = (0.5 * quad / var + 0.5 * n_in * (abs(n * np.log(var)) + abs(2.0 * sld))
                         + n_in * n / 2 * np.log(2 * np.pi) + abs(det))
    return (out, mag) if return_scale else out


# --------------------------------------------------------------------------
# fit / predict  (reference: gsum/models.py:671-738, 753-845)
# --------------------------------------------------------------------------

def cgp_fit(kernel, X, y, center=0, disp=0, df=1, scale=1, sd=None, nugget=1e-10):
    """ConjugateGaussianProcess.fit with a fixed kernel (optimizer inactive).

    Follows models.py:705-737.  Returns the fitted attributes as a dict.
    """
    center0, disp0, df0, scale0 = _priors(center, disp, df, scale, sd)
    basis = _ones_basis(X)                                     # models.py:705
    lml_value = cgp_lml(kernel, None, X, y, center, disp, df, scale, sd, nugget)  # :668-669
    corr = kernel(X)                                           # models.py:708
    L = cholesky(corr + nugget * np.eye(len(X)))               # models.py:711
    center_n = posterior_center(y, L, basis, center0, disp0)   # models.py:721
    disp_n = posterior_disp(y, L, basis, disp0)                # models.py:725
    df_n = posterior_df(y, df0)                                # models.py:729
    scale_sq = posterior_scale_sq(y, L, basis, center0, disp0, df0, scale0)  # :730
    return dict(kernel=kernel, X=X, y=y, nugget=nugget, corr=corr, corr_L=L,
                center=center_n, disp=disp_n, df=df_n, scale=np.sqrt(scale_sq),
                cov_factor=cov_factor(scale_sq, df_n),         # models.py:735-736
                lml=lml_value)


def cgp_predict(fit, Xnew, return_std=False, return_cov=False, Xc=None, y=None,
                pred_noise=False):
    """ConjugateGaussianProcess.predict on a fitted state.  models.py:789-845."""
    if return_std and return_cov:
        raise RuntimeError("Only one of return_std or return_cov may be True")
    kernel, nugget = fit["kernel"], fit["nugget"]
    if Xc is None:
        Xc, L = fit["X"], fit["corr_L"]                        # models.py:797-800
    else:
        L = cholesky(kernel(Xc) + nugget * np.eye(len(Xc)))    # models.py:807-809
    if y is None:
        y = fit["y"]
    m_old = _ones_basis(Xc) @ fit["center"]                    # models.py:818
    m_new = _ones_basis(Xnew) @ fit["center"]                  # models.py:819
    R_on = kernel(Xc, Xnew)                                    # models.py:822
    R_no = R_on.T
    R_nn = kernel(Xnew)                                        # models.py:824
    if y.ndim == 1:
        y = y[:, None]
    alpha = _rsolve(L, y - m_old[:, None])                     # models.py:831
    m_pred = np.squeeze(m_new[:, None] + R_no @ alpha)         # models.py:832
    if return_std or return_cov:
        R_pred = R_nn - R_no @ _rsolve(L, R_on)                # models.py:836
        if pred_noise:
            R_pred += nugget * np.eye(len(Xnew))               # models.py:837-838
        var = cov_factor(fit["scale"] ** 2, fit["df"])         # models.py:840
        K_pred = np.squeeze(var * R_pred)
        if return_std:
            return m_pred, np.sqrt(np.diag(K_pred))            # models.py:842-843
        return m_pred, K_pred
    return m_pred


# --------------------------------------------------------------------------
# truncation-layer mean / cov / predict(kind='trunc')  (reference: gsum/models.py:1337-1365, 1456-1477)
# --------------------------------------------------------------------------

def _vec(v, X):
    """ratio / ref at the points X: a callable (models.py:1309-1317 wraps scalars into one), a scalar or an array."""
    if callable(v):
        return np.asarray(v(X))
    n = X.shape[0]
    return v * np.ones(n) if np.ndim(v) == 0 else np.asarray(v)


def trunc_mean(center, X, ratio, ref, start=0, end=np.inf, excluded=None):
    """ref(X) * geometric_sum(ratio(X)) * (basis @ center).  models.py:1337-1340."""
    coeff_mean = _ones_basis(X) @ np.atleast_1d(center)
    return _vec(ref, X) * geometric_sum(_vec(ratio, X), start, end, excluded) * coeff_mean


def trunc_cov(factor, kernel, X, Xp=None, ratio=0.5, ref=1, start=0, end=np.inf, excluded=None):
    """ref_mat * geometric_sum(ratio_mat) * cov_factor * kernel(X, Xp).  models.py:1342-1348, 599."""
    coeff_cov = factor * kernel(X, Xp)
    Xp = X if Xp is None else Xp                    # reassigned after the kernel call (models.py:1344)
    ratio_mat = _vec(ratio, X)[:, None] * _vec(ratio, Xp)
    ref_mat = _vec(ref, X)[:, None] * _vec(ref, Xp)
    return ref_mat * geometric_sum(ratio_mat, start, end, excluded) * coeff_cov


def trunc_basis(X, ratio, ref, start=0, end=np.inf, excluded=None):
    """models.py:1350-1354."""
    return _vec(ref, X)[:, None] * geometric_sum(_vec(ratio, X)[:, None], start, end, excluded) * _ones_basis(X)


def trunc_predict_trunc(center, factor, kernel, X, order, ratio, ref, excluded=None, fitted=True,
                        return_std=False, return_cov=False):
    """TruncationProcess.predict(kind='trunc') without constraints (fitted=True: models.py:1460-1461,
    1474-1483, covariance built with Xp = X given explicitly, so no WhiteKernel term) and the unfitted path
    underlying_properties (fitted=False: models.py:1356-1365, one-argument covariance)."""
    m = trunc_mean(center, X, ratio, ref, start=order + 1, end=np.inf, excluded=excluded)
    if not (return_std or return_cov):
        return m
    K = trunc_cov(factor, kernel, X, X if fitted else None, ratio, ref, start=order + 1, end=np.inf,
                  excluded=excluded)
    return (m, K) if return_cov else (m, np.sqrt(np.diag(K)))


def trunc_predict(center, factor, kernel, X, order, ratio, ref, Xc, y, excluded=None, kind="both", dX=None, dy=None,
                  return_std=False, return_cov=False):
    """Fitted TruncationProcess.predict for every kind (models.py:1430-1483): the interpolation block conditions
    y_order on (Xc, y) with covariances over orders 0..order, the truncation block adds the orders beyond, itself
    conditioned on (dX, dy) when constraints were given to fit.  Conditioning uses numpy.linalg.solve (LU) on the
    un-jittered cov(Xc, Xc), exactly as the reference (:1449, 1452, 1470, 1473)."""
    if kind not in ("both", "interp", "trunc"):
        raise ValueError('kind must be one of "both", "interp" or "trunc"')
    want = return_std or return_cov
    m_pred, K_pred = 0, 0

    def block(Xo, resid_from, start, end):
        m_old = trunc_mean(center, Xo, ratio, ref, start=start, end=end, excluded=excluded)
        m_new = trunc_mean(center, X, ratio, ref, start=start, end=end, excluded=excluded)
        K_oo = trunc_cov(factor, kernel, Xo, Xo, ratio, ref, start=start, end=end, excluded=excluded)
        K_on = trunc_cov(factor, kernel, Xo, X, ratio, ref, start=start, end=end, excluded=excluded)
        K_nn = trunc_cov(factor, kernel, X, X, ratio, ref, start=start, end=end, excluded=excluded)
        m = m_new + K_on.T @ np.linalg.solve(K_oo, resid_from - m_old)
        K = K_nn - K_on.T @ np.linalg.solve(K_oo, K_on) if want else 0
        return m, K

    if kind in ("both", "interp"):
        m, K = block(Xc, y, 0, order)
        m_pred, K_pred = m_pred + m, K_pred + K
    if kind in ("both", "trunc"):
        if dX is not None:
            m, K = block(dX, dy, order + 1, np.inf)
        else:
            m = trunc_mean(center, X, ratio, ref, start=order + 1, end=np.inf, excluded=excluded)
            K = trunc_cov(factor, kernel, X, X, ratio, ref, start=order + 1, end=np.inf, excluded=excluded) if want else 0
        m_pred, K_pred = m_pred + m, K_pred + K
    if return_cov:
        return m_pred, K_pred
    if return_std:
        return m_pred, np.sqrt(np.diag(K_pred))
    return m_pred


def cgp_prior_predict(kernel, Xnew, center=0, df=1, scale=1, sd=None, return_std=False, return_cov=False):
    """Unfitted ConjugateGaussianProcess.predict = underlying_properties.  models.py:740-749, 792-793, 587-599."""
    center0, _, df0, scale0 = _priors(center, 0, df, scale, sd)
    mean = _ones_basis(Xnew) @ center0
    if not (return_std or return_cov):
        return mean
    if df0 <= 2:
        raise ValueError("df must be greater than 2 for the covariance to exist")
    cov = cov_factor(scale0 ** 2, df0) * kernel(Xnew)
    return (mean, cov) if return_cov else (mean, np.sqrt(np.diag(cov)))


# --------------------------------------------------------------------------
# Student-t process  (reference: gsum/models.py:1091-1273) and TruncationTP (:1519-1570)
# --------------------------------------------------------------------------

def csp_lml(kernel, theta, X, y, center=0, disp=0, df=1, scale=1, sd=None, nugget=1e-10):
    """ConjugateStudentProcess.log_marginal_likelihood, value path (models.py:1186-1259, cholesky)."""
    from scipy.special import loggamma
    center0, disp0, df0, scale0 = _priors(center, disp, df, scale, sd)
    if theta is not None:
        kernel = kernel.clone_with_theta(np.asarray(theta, dtype=float))      # :1202
    ny = _num_y(y)                                                            # :1192
    R = kernel(X)                                                             # :1206
    R[np.diag_indices_from(R)] += nugget                                      # :1208
    N = R.shape[0]
    try:
        L = cholesky(R)                                                       # :1215
    except np.linalg.LinAlgError:
        return -np.inf
    dfn = posterior_df(y, df0)                                                # :1225
    basis = _ones_basis(X)
    disp_n = posterior_disp(y, L, basis, disp0)                               # :1238
    scale_sq = posterior_scale_sq(y if y.ndim > 1 else y[:, None], L, basis, center0, disp0, df0, scale0)   # :1239-1242
    scale_n = np.sqrt(scale_sq)                                               # :1244

    def log_norm(df_, scale_, disp_):                                         # :1246-1252
        norm = loggamma(df_ / 2.) - df_ / 2. * np.log(df_ * scale_ ** 2 / 2.)
        log_det = np.linalg.slogdet(2 * np.pi * disp_)[1]
        if log_det != -np.inf:
            norm += 0.5 * log_det
        return norm

    logdet_R = 2 * np.log(np.diag(L)).sum()                                   # :1255
    return log_norm(dfn, scale_n, disp_n) - log_norm(df0, scale0, disp0) \
        - ny / 2. * (N * np.log(2 * np.pi) + logdet_R)                         # :1262-1263


def csp_cov(kernel, X, Xp, df, scale, disp):
    """ConjugateStudentProcess.cov: var * (kernel(X, Xp) + basis disp basis^T).  models.py:1099-1125."""
    if df <= 2:
        raise ValueError("df must be greater than 2 for the covariance to exist")
    corr = kernel(X, Xp)
    Xp = X if Xp is None else Xp
    return cov_factor(scale ** 2, df) * (corr + _ones_basis(X) @ np.atleast_2d(disp) @ _ones_basis(Xp).T)


def csp_predict(fit, Xnew, return_std=False, return_cov=False, Xc=None, y=None):
    """Fitted ConjugateStudentProcess.predict = the shared predict plus the covariance from integrating out the
    mean, var * b disp b^T with the conditional basis b = basis_new - R_no R^-1 basis_old.  models.py:1127-1184.
    (Standard deviations are ADDED, as the reference does at :1177.)"""
    pred = cgp_predict(fit, Xnew, return_std=return_std, return_cov=return_cov, Xc=Xc, y=y)
    kernel, nugget = fit["kernel"], fit["nugget"]
    if Xc is None:
        Xc, L = fit["X"], fit["corr_L"]
    else:
        L = cholesky(kernel(Xc) + nugget * np.eye(len(Xc)))
    R_no = kernel(Xnew, Xc)
    basis = _ones_basis(Xnew) - R_no @ _rsolve(L, _ones_basis(Xc))           # :1168
    mean_cov = fit["cov_factor"] * (basis @ fit["disp"] @ basis.T)           # :1174
    if return_std:
        return pred[0], pred[1] + np.sqrt(np.diag(mean_cov))
    if return_cov:
        return pred[0], pred[1] + mean_cov
    return pred


def csp_prior_predict(kernel, Xnew, center=0, disp=0, df=1, scale=1, sd=None, return_std=False, return_cov=False):
    """Unfitted ConjugateStudentProcess.predict: underlying_properties with the Student covariance, then the
    basis term once more (models.py:1138-1145, 1174-1183)."""
    center0, disp0, df0, scale0 = _priors(center, disp, df, scale, sd)
    mean = _ones_basis(Xnew) @ center0
    if not (return_std or return_cov):
        return mean
    cov = csp_cov(kernel, Xnew, None, df0, scale0, disp0)
    b = _ones_basis(Xnew)
    mean_cov = cov_factor(scale0 ** 2, df0) * (b @ disp0 @ b.T)
    if return_std:
        return mean, np.sqrt(np.diag(cov)) + np.sqrt(np.diag(mean_cov))
    return mean, cov + mean_cov


def ttp_predict(fit, X, order, ratio, ref, Xc, y, excluded=None, kind="both", dX=None, dy=None, return_std=False,
                return_cov=False):
    """TruncationTP.predict (models.py:1527-1570).  The Gaussian part is TruncationProcess.predict with kind='both'
    whatever ``kind`` says (it is not forwarded, :1528-1531) and with the Student covariance var (kernel + disp);
    ``kind`` only selects which conditional bases enter the rank-one term."""
    var, disp = fit["cov_factor"], fit["disp"]
    kern_t = _PlusConstant(fit["kernel"], float(disp[0, 0]))
    factor_t = cov_factor(fit["scale"] ** 2, fit["df"])
    pred = trunc_predict(fit["center"], factor_t, kern_t, X, order, ratio, ref, Xc, y, excluded=excluded, kind="both",
                         dX=dX, dy=dy, return_std=return_std, return_cov=return_cov)
    if not return_std and not return_cov:
        return pred
    m = X.shape[0]
    basis_lower, basis_trunc = np.zeros((m, 1)), np.zeros((m, 1))
    if kind in ("both", "interp"):
        K_oo = trunc_cov(factor_t, kern_t, Xc, Xc, ratio, ref, 0, order, excluded)
        K_no = trunc_cov(factor_t, kern_t, X, Xc, ratio, ref, 0, order, excluded)
        basis_lower = trunc_basis(X, ratio, ref, 0, order, excluded) \
            - K_no @ np.linalg.solve(K_oo, trunc_basis(Xc, ratio, ref, 0, order, excluded))
    if kind in ("both", "trunc"):
        if dX is not None:
            K_oo = trunc_cov(factor_t, kern_t, dX, dX, ratio, ref, order + 1, np.inf, excluded)
            K_no = trunc_cov(factor_t, kern_t, X, dX, ratio, ref, order + 1, np.inf, excluded)
            basis_trunc = trunc_basis(X, ratio, ref, order + 1, np.inf, excluded) \
                - K_no @ np.linalg.solve(K_oo, trunc_basis(dX, ratio, ref, order + 1, np.inf, excluded))
        else:
            basis_trunc = trunc_basis(X, ratio, ref, order + 1, np.inf, excluded)
    b = basis_lower + basis_trunc
    mean_cov = var * b @ disp @ b.T                                           # :1564
    if return_std:
        return pred[0], pred[1] + np.sqrt(np.diag(mean_cov))
    return pred[0], pred[1] + mean_cov


class _PlusConstant:
    """kernel(X, Xp) + c: the Student process's corr + basis disp basis^T for the constant basis."""

    def __init__(self, kernel, c):
        self.kernel, self.c = kernel, c

    def __call__(self, X, Xp=None):
        return self.kernel(X, Xp) + self.c


# --------------------------------------------------------------------------
# gradient of the log marginal likelihood w.r.t. the kernel's log-hyperparameters
# (reference: gsum/models.py:957-958, 989-999, 1022-1024, 1041-1056; helpers at :221-232, 271-279, 447-457)
# --------------------------------------------------------------------------

def _hyper_gradients(y, L, basis, dR, center0, disp0, df0, scale0):
    """(center, d_center, disp, d_disp, scale_sq, d_scale_sq) as compute_center / compute_disp / compute_scale_sq
    return them with eval_gradient=True."""
    ny = _num_y(y)
    avg = _avg_y(y)
    P = dR.shape[-1]
    disp = posterior_disp(y, L, basis, disp0)
    center = posterior_center(y, L, basis, center0, disp0)
    invR_basis = _rsolve(L, basis)
    if np.all(disp0 == 0):                                                     # :203-204, 262-263
        d_center = np.zeros((*center0.shape, P))
        d_disp = np.zeros((*disp0.shape, P))
    else:
        invR_diff = _rsolve(L, basis @ center - avg)                           # :226
        d_center = ny * disp @ np.einsum('ji,jkp,k->ip', invR_basis, dR, invR_diff)      # :229
        invRBV = invR_basis @ disp                                             # :274
        d_disp = ny * np.einsum('ji,jkp,kl->ilp', invRBV, dR, invRBV)          # :276
    scale_sq = posterior_scale_sq(y, L, basis, center0, disp0, df0, scale0)
    if df0 == np.inf:
        d_scale_sq = np.zeros(P)                                               # :419-421
    else:
        yc = y - avg[:, None]
        invR_yc = _rsolve(L, yc)                                               # :429
        avg_c = avg - basis @ center0                                          # :431
        N = len(avg)
        mat = np.eye(N) - ny * invR_basis @ disp @ basis.T                     # :439
        m = ny * mat @ _rsolve(L, avg_c)                                       # :440
        d_scale_sq = -np.einsum('ji,jkp,ki->p', invR_yc, dR, invR_yc)          # :453
        d_scale_sq -= np.einsum('i,ijp,j->p', m, dR, m) / ny                   # :454
        d_scale_sq /= posterior_df(y, df0)                                     # :455
    return center, d_center, disp, d_disp, scale_sq, d_scale_sq


def cgp_lml_grad(kernel, theta, X, y, center=0, disp=0, df=1, scale=1, sd=None, nugget=1e-10):
    """ConjugateGaussianProcess.log_marginal_likelihood(theta, eval_gradient=True): (value, gradient)."""
    center0, disp0, df0, scale0 = _priors(center, disp, df, scale, sd)
    kernel = kernel.clone_with_theta(np.asarray(theta, dtype=float))           # :953
    R, dR = kernel(X, eval_gradient=True)                                      # :958
    R[np.diag_indices_from(R)] += nugget                                       # :963
    try:
        L = cholesky(R)
    except np.linalg.LinAlgError:
        return -np.inf, np.zeros_like(theta)                                   # :970-972
    if y.ndim == 1:
        y = y[:, np.newaxis]
    basis = _ones_basis(X)
    dfn = posterior_df(y, df0)
    center_n, d_center, _, _, scale2, d_scale2 = _hyper_gradients(y, L, basis, dR, center0, disp0, df0, scale0)
    grad_var = cov_factor(d_scale2, dfn)                                       # :998
    grad_mean = basis @ d_center                                               # :999
    mean = basis @ center_n
    var = cov_factor(scale2, dfn)
    Lk = np.sqrt(var) * L
    K_gradient = var * dR + grad_var * R[:, :, None]                           # :1024
    y_train = y - mean[:, None]
    N = R.shape[0]
    alpha = cho_solve((Lk, True), y_train)
    ll = -0.5 * np.einsum("ik,ik->k", y_train, alpha) - np.log(np.diag(Lk)).sum() - N / 2 * np.log(2 * np.pi)
    tmp = np.einsum("ik,jk->ijk", alpha, alpha)                                # :1042
    tmp -= cho_solve((Lk, True), np.eye(N))[:, :, np.newaxis]                  # :1044
    g = 0.5 * np.einsum("ijl,ijk->kl", tmp, K_gradient)                        # :1049
    g -= grad_mean.T @ alpha                                                   # :1052
    return ll.sum(-1), g.sum(-1)                                               # :1055


def csp_lml_grad(kernel, theta, X, y, center=0, disp=0, df=1, scale=1, sd=None, nugget=1e-10):
    """ConjugateStudentProcess.log_marginal_likelihood(theta, eval_gradient=True) as written at models.py:1227-1236,
    1264-1272.  (The reference calls ``kernel(X, eval_gradient)`` at :1204, which hands True to the Y argument and
    raises; the evident intent ``kernel(X, eval_gradient=True)`` is used.  Pinned by finite differences of the
    reference's value path in tests/golden/gradient.json.)"""
    center0, disp0, df0, scale0 = _priors(center, disp, df, scale, sd)
    value = csp_lml(kernel, theta, X, y, center, disp, df, scale, sd, nugget)
    kernel = kernel.clone_with_theta(np.asarray(theta, dtype=float))
    R, dR = kernel(X, eval_gradient=True)
    R[np.diag_indices_from(R)] += nugget
    L = cholesky(R)
    ny = _num_y(y)
    y2 = y if y.ndim > 1 else y[:, None]
    basis = _ones_basis(X)
    dfn = posterior_df(y, df0)
    _, _, disp_n, d_disp, scale_sq, d_scale_sq = _hyper_gradients(y2, L, basis, dR, center0, disp0, df0, scale0)
    g = -(ny / 2.) * np.trace(cho_solve((L, True), dR.reshape(len(X), -1)).reshape(dR.shape), axis1=0, axis2=1)   # :1266
    g = g - (dfn / 2.) * d_scale_sq / scale_sq                                 # :1269
    if not np.all(disp_n == 0):
        g = g + 0.5 * np.einsum('ij,ijp->p', np.linalg.inv(disp_n), d_disp)    # :1271-1272
    return value, g
