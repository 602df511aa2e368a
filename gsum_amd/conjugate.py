"""ConjugateGaussianProcess on MI355X — host-side mirror of gsum/models.py:31-1057.

The class keeps the reference's constructor, ``fit`` / ``predict`` / ``log_marginal_likelihood``
/ ``mean`` / ``cov`` surface and fitted attributes, but every O(n^2)/O(n^3) operation is one call
into libgsum_hip.so:

    kernel(X)                     -> gsum_kernel_build[_dev]      (models.py:708, 958-960)
    numpy.linalg.cholesky         -> gsum_potrf_lower             (models.py:711, 809, 969)
    4x cho_solve + traces + N x N
    Woodbury temporary            -> one Gram matrix G = W^T W, W = L^-1 [Y | B], plus sum(log diag L)
                                     (models.py:432-445, 1015, 1032-1035)

What is left on the host is O((n_curves+1)^2) scalar algebra on G (``posterior_from_gram``).
There is no CPU fallback; an unsupported kernel / option raises.
"""
from __future__ import annotations

import os
import warnings

import numpy as np
from scipy.linalg import cho_solve, inv
from scipy.optimize import fmin_l_bfgs_b
from scipy.special import loggamma
from sklearn.base import clone
from sklearn.exceptions import ConvergenceWarning
from sklearn.utils import check_random_state

from ._lib import GSUM_MAX_RHS, default_context
from .kernels import default_kernel, describe_gradient, describe_gradients, describe_kernel, describe_thetas

__all__ = ["ConjugateGaussianProcess", "ConjugateStudentProcess", "posterior_from_gram", "lml_from_gram",
           "lml_from_gram_batch", "student_lml_from_gram", "hyper_gradients_from_gram", "lml_grad_from_gram",
           "student_lml_grad_from_gram", "cov_factor"]


# ---------------------------------------------------------------------------------------------
# host algebra on the Gram matrix, p = 1 basis column
# ---------------------------------------------------------------------------------------------

def cov_factor(scale_sq, df):
    """sigma^2 = nu tau^2 / (nu - 2); tau^2 when nu = inf (models.py:490-503)."""
    if df != np.inf:
        return df * scale_sq / (df - 2)
    return scale_sq


def posterior_from_gram(G, n_points, center0, disp0, df0, scale0):
    """Conjugate updates from G = [Y | 1]^T R^-1 [Y | 1].

    Restates compute_center / compute_disp / compute_df / compute_scale_sq (models.py:170-457) in
    terms of the (ny+1) x (ny+1) Gram matrix, so no further solve is needed:
      q = ybar^T R^-1 ybar, b = 1^T R^-1 ybar, g = 1^T R^-1 1 are bilinear forms of G.
    Returns dict(center (1,), disp (1,1), df, scale_sq, cov_factor, S) with
    S = sum_k (y_k - eta)^T R^-1 (y_k - eta).
    """
    G = np.asarray(G, dtype=float)
    ny = G.shape[0] - 1
    Gyy, gyb, g = G[:ny, :ny], G[:ny, ny], G[ny, ny]
    center0 = np.atleast_1d(np.asarray(center0, dtype=float))
    disp0 = np.atleast_2d(np.asarray(disp0, dtype=float))
    if center0.shape != (1,) or disp0.shape != (1, 1):
        raise ValueError("center must be a scalar and disp a scalar (single constant basis function)")
    eta0, V0 = center0[0], disp0[0, 0]
    w = np.full(ny, 1.0 / ny)
    q = w @ Gyy @ w
    b = gyb @ w
    tr = np.trace(Gyy)
    if V0 == 0:                                   # models.py:201-206, 260-265
        V, eta = 0.0, eta0
    else:
        V = 1.0 / (1.0 / V0 + ny * g)             # models.py:269-270
        eta = V * (eta0 / V0 + ny * b)            # models.py:219-220
    df = df0 + n_points * ny                      # models.py:302
    if df0 == np.inf:
        scale_sq = scale0 ** 2                    # models.py:419-422
    else:
        quad = tr - ny * q                        # models.py:430-433
        a = q - 2.0 * eta0 * b + eta0 * eta0 * g  # (ybar - eta0)^T R^-1 (ybar - eta0)
        v = b - g * eta0
        quad2 = ny * (a - ny * v * V * v)         # models.py:435-445 (Woodbury term, no N x N matrix)
        scale_sq = (df0 * scale0 ** 2 + quad + quad2) / df   # models.py:448
    S = tr - 2.0 * ny * eta * b + ny * eta * eta * g
    return dict(center=np.array([eta]), disp=np.array([[V]]), df=df, scale_sq=scale_sq,
                cov_factor=cov_factor(scale_sq, df), S=S)


def lml_from_gram(G, sum_log_diag, n_points, center0, disp0, df0, scale0):
    """log marginal likelihood from (G, sum log diag L).  models.py:1007-1039."""
    post = posterior_from_gram(G, n_points, center0, disp0, df0, scale0)
    ny = np.asarray(G).shape[0] - 1
    var = post["cov_factor"]
    logdet_K = n_points * np.log(var) + 2.0 * sum_log_diag          # models.py:1014-1015
    lml = -0.5 * post["S"] / var - 0.5 * ny * logdet_K - ny * n_points / 2.0 * np.log(2.0 * np.pi)
    return float(lml), post


def lml_from_gram_batch(G, sum_log_diag, n_points, center0, disp0, df0, scale0):
    """Vectorised :func:`lml_from_gram` over a stack of Gram matrices ``G[b]`` (same formulas, numpy
    broadcasting instead of a Python loop: a grid row is hundreds of evaluations)."""
    G = np.asarray(G, dtype=float)
    s = np.asarray(sum_log_diag, dtype=float)
    ny = G.shape[-1] - 1
    center0 = np.atleast_1d(np.asarray(center0, dtype=float))
    disp0 = np.atleast_2d(np.asarray(disp0, dtype=float))
    if center0.shape != (1,) or disp0.shape != (1, 1):
        raise ValueError("center must be a scalar and disp a scalar (single constant basis function)")
    eta0, V0 = center0[0], disp0[0, 0]
    Gyy, gyb, g = G[:, :ny, :ny], G[:, :ny, ny], G[:, ny, ny]
    q = Gyy.sum(axis=(1, 2)) / (ny * ny)
    b = gyb.sum(axis=1) / ny
    tr = np.trace(Gyy, axis1=1, axis2=2)
    if V0 == 0:
        V, eta = np.zeros_like(g), np.full_like(g, eta0)
    else:
        V = 1.0 / (1.0 / V0 + ny * g)
        eta = V * (eta0 / V0 + ny * b)
    df = df0 + n_points * ny
    if df0 == np.inf:
        scale_sq = np.broadcast_to(np.asarray(scale0, dtype=float) ** 2, g.shape)       # scale0 may be one value per entry
    else:
        quad = tr - ny * q
        a = q - 2.0 * eta0 * b + eta0 * eta0 * g
        v = b - g * eta0
        scale_sq = (df0 * scale0 ** 2 + quad + ny * (a - ny * v * V * v)) / df
    var = scale_sq if df == np.inf else df * scale_sq / (df - 2)
    S = tr - 2.0 * ny * eta * b + ny * eta * eta * g
    logdet_K = n_points * np.log(var) + 2.0 * s
    return -0.5 * S / var - 0.5 * ny * logdet_K - ny * n_points / 2.0 * np.log(2.0 * np.pi)


def student_lml_from_gram(G, sum_log_diag, n_points, center0, disp0, df0, scale0):
    """ConjugateStudentProcess.log_marginal_likelihood from (G, sum log diag L): the ratio of the normalisation
    constants of the normal scaled-inverse-chi-squared posterior and prior.  models.py:1186-1259 (value path)."""
    post = posterior_from_gram(G, n_points, center0, disp0, df0, scale0)
    ny = np.asarray(G).shape[0] - 1
    disp0 = np.atleast_2d(np.asarray(disp0, dtype=float))

    def log_norm(df_, scale_, disp_):                                   # models.py:1234-1240
        norm = loggamma(df_ / 2.) - df_ / 2. * np.log(df_ * scale_ ** 2 / 2.)
        log_det = np.linalg.slogdet(2 * np.pi * disp_)[1]
        if log_det != -np.inf:
            norm += 0.5 * log_det
        return norm

    logdet_R = 2.0 * sum_log_diag                                       # models.py:1243
    lml = log_norm(post["df"], np.sqrt(post["scale_sq"]), post["disp"]) - log_norm(df0, scale0, disp0) \
        - ny / 2. * (n_points * np.log(2 * np.pi) + logdet_R)           # models.py:1250-1251
    return float(lml), post


# ---------------------------------------------------------------------------------------------
# gradients with respect to the kernel's log-hyperparameters (SURVEY.md kernel K6)
#
# Every n-vector the reference contracts with dR_p lies in the span of the columns of V = R^-1 [Y | 1], so each
# einsum('..,jkp,..') at models.py:229, 276, 453-454, 1049 is a bilinear form of H_p = V^T dR_p V (from the device),
# and the one genuinely n x n contraction is trace_p = tr(R^-1 dR_p).
# ---------------------------------------------------------------------------------------------

def hyper_gradients_from_gram(G, H, post, center0, disp0, df0):
    """d center / d theta, d disp / d theta, d scale^2 / d theta (each length P) for the constant basis.
    models.py:221-232 (compute_center), 271-279 (compute_disp), 447-457 (compute_scale_sq, Woodbury form)."""
    G = np.asarray(G, dtype=float)
    H = np.asarray(H, dtype=float)
    ny = G.shape[0] - 1
    P = H.shape[0]
    eta0 = float(np.atleast_1d(center0)[0])
    V0 = float(np.atleast_2d(disp0)[0, 0])
    center, disp, df = float(post["center"][0]), float(post["disp"][0, 0]), post["df"]
    w = np.full(ny, 1.0 / ny)
    e1 = np.zeros(ny + 1)
    e1[ny] = 1.0
    if V0 == 0:                                                   # models.py:203-204, 262-263
        d_center, d_disp = np.zeros(P), np.zeros(P)
    else:
        a_diff = np.append(-w, center)                            # R^-1 (basis center - y_avg) = V a_diff   (:226)
        d_center = ny * disp * np.array([H[p][ny] @ a_diff for p in range(P)])        # :229
        d_disp = ny * disp * disp * H[:, ny, ny]                  # :275-276
    if df0 == np.inf:
        d_scale_sq = np.zeros(P)                                  # models.py:419-421
    else:
        Cc = np.eye(ny + 1)[:, :ny] - np.outer(np.append(w, 0.0), np.ones(ny))        # y - y_avg = Z Cc      (:427)
        a_c = np.append(w, -eta0)                                 # y_avg - basis center0 = Z a_c             (:431)
        s = G[ny] @ a_c                                           # basis^T R^-1 (y_avg - basis center0)
        a_m = ny * (a_c - ny * disp * s * e1)                     # mat_invR_avg_yc = V a_m                   (:439-440)
        d_scale_sq = np.array([-np.trace(Cc.T @ H[p] @ Cc) - (a_m @ H[p] @ a_m) / ny for p in range(P)]) / df   # :453-455
    return d_center, d_disp, d_scale_sq


def lml_grad_from_gram(G, sum_log_diag, trace, H, n_points, center0, disp0, df0, scale0):
    """ConjugateGaussianProcess.log_marginal_likelihood(theta, eval_gradient=True) from device pieces: (lml, grad).
    models.py:989-999, 1022-1024, 1041-1056."""
    lml, post = lml_from_gram(G, sum_log_diag, n_points, center0, disp0, df0, scale0)
    G = np.asarray(G, dtype=float)
    H = np.asarray(H, dtype=float)
    trace = np.asarray(trace, dtype=float)
    ny = G.shape[0] - 1
    d_center, _, d_scale_sq = hyper_gradients_from_gram(G, H, post, center0, disp0, df0)
    df, var = post["df"], post["cov_factor"]
    grad_var = cov_factor(d_scale_sq, df)                         # models.py:998
    center = float(post["center"][0])
    Cy = np.eye(ny + 1)[:, :ny].copy()
    Cy[ny, :] = -center                                           # y - mean = Z Cy                            (:1026)
    quad_G = np.trace(Cy.T @ G @ Cy)                              # sum_l (y_l - m)^T R^-1 (y_l - m)
    one_G = (G[ny] @ Cy).sum()                                    # sum_l basis^T R^-1 (y_l - m)
    quad_H = np.array([np.trace(Cy.T @ Hp @ Cy) for Hp in H])
    # K = var R, K_gradient = var dR + grad_var R (:1022-1024); alpha_l = V Cy[:, l] / var
    grad = 0.5 * (var * quad_H + grad_var * quad_G) / var ** 2 \
        - 0.5 * ny * (trace + n_points * grad_var / var) \
        - d_center * one_G / var                                  # :1049, 1052
    return lml, grad


def student_lml_grad_from_gram(G, sum_log_diag, trace, H, n_points, center0, disp0, df0, scale0):
    """ConjugateStudentProcess.log_marginal_likelihood(theta, eval_gradient=True): models.py:1227-1236, 1264-1271."""
    lml, post = student_lml_from_gram(G, sum_log_diag, n_points, center0, disp0, df0, scale0)
    ny = np.asarray(G).shape[0] - 1
    _, d_disp, d_scale_sq = hyper_gradients_from_gram(G, H, post, center0, disp0, df0)
    grad = -(ny / 2.) * np.asarray(trace, dtype=float)            # :1266
    grad = grad - (post["df"] / 2.) * d_scale_sq / post["scale_sq"]                   # :1269
    disp = float(post["disp"][0, 0])
    if disp != 0:
        grad = grad + 0.5 * d_disp / disp                         # :1271-1272
    return lml, grad


# ---------------------------------------------------------------------------------------------
# the classes
# ---------------------------------------------------------------------------------------------

class ConjugateGaussianProcess:
    """Conjugate-prior GP; same constructor and methods as gsum.ConjugateGaussianProcess.

    Parameters are those of gsum/models.py:107-109.  Additive: ``device`` (GPU index; default
    ``$LOCAL_RANK`` or 0) and ``backend`` ('hip', the default, or 'cpu': the same operator interface on
    numpy / scipy; also ``GSUM_BACKEND``).  ``basis`` other than ``None`` raises ``NotImplementedError`` (the reference
    does not support it either, models.py:149-150).  ``decomposition='eig'`` (models.py:713-717, 810-811, 973-974) is accepted: every quantity
    the reference computes through ``(eig, Q) = eigh(R)`` -- R^-1 y, log det R, the predictive pieces -- is the same quantity the factorisation
    on the device gives (the reference's own two modes agree to 1e-14), so likelihood, fit and predict run the one device path; the
    eigen-decomposition itself exists only as the attributes ``_eigh_tuple_`` / ``corr_sqrt_`` / ``corr_L_`` (= Q sqrt(eig), models.py:715-717),
    computed from ``corr_`` on the host when one of them is read.
    """

    def __init__(self, kernel=None, center=0, disp=0, df=1, scale=1, sd=None, basis=None, nugget=1e-10,
                 optimizer='fmin_l_bfgs_b', n_restarts_optimizer=0, copy_X_train=True, random_state=None,
                 decomposition='cholesky', device=None, backend=None):
        self.kernel = kernel
        self._center_0 = np.atleast_1d(center)
        self._disp_0 = np.atleast_2d(disp)
        if sd is not None:                      # models.py:115-117
            self._df_0 = np.inf
            self._scale_0 = sd
        else:
            self._df_0 = df
            self._scale_0 = scale
        self._fit = False
        self.X_train_ = None
        self.y_train_ = None
        self.center_ = None
        self.disp_ = None
        self.df_ = None
        self.scale_ = None
        self.cov_factor_ = self.cbar_sq_mean_ = None
        self.kernel_ = None
        self._rng = None
        self.nugget = nugget
        self.copy_X_train = copy_X_train
        self.random_state = random_state
        self.n_restarts_optimizer = n_restarts_optimizer
        self.optimizer = optimizer
        self.decomposition = decomposition
        self._default_kernel = default_kernel()
        if basis is not None:
            # the reference only ever assigns self.basis when basis is None (models.py:149-150)
            raise NotImplementedError("only the constant basis (basis=None) is supported")
        self.basis = lambda X: np.ones((np.shape(X)[0], 1))
        self.basis_train_ = None
        self.device = device
        # 'hip' (default; also through GSUM_BACKEND): libgsum_hip.so on an MI355X, loud failure without it.  'cpu': the same
        # operator interface on numpy / scipy / scikit-learn (gsum_amd/_cpu.py; BASELINE config 1) -- only when asked for
        self.backend = backend if backend is not None else os.environ.get("GSUM_BACKEND", "hip")
        if self.backend not in ("hip", "cpu"):
            raise ValueError("backend must be 'hip' or 'cpu'")
        self.batch_restarts = True   # multi-start fits advance in lock step, objective evaluations batched on the device
        self._ctx = None
        self._L_dev = None          # device-resident Cholesky factor of kernel_(X_train_) + nugget
        self._replicas = {}         # ... and its copies on the other devices of predict(devices=...) (id(context) -> factor)
        self._corr = None
        self._corr_L = None
        self._gram = None

    # -- the reference's classmethod / operator surface (models.py:170-503, 601-628) ---------------------------------
    # Host utilities on a caller-supplied square root of R, with the reference's signatures.  The classes here never
    # call them (fit / log_marginal_likelihood read everything off the device's Gram matrix); they are kept so that
    # user code written against gsum's public surface keeps working.  ``sqrt_R`` is a host array as in the reference
    # or a device factor (``DeviceMatrix``), in which case the solves run on the GPU (gsum_cho_solve).  All of them
    # reduce to one solve V = R^-1 [y | basis] and the (n_curves + p) x (n_curves + p) Gram matrix [y | basis]^T V.
    @staticmethod
    def num_y(y):                                                  # models.py:601-607
        return y.shape[1] if np.ndim(y) == 2 else 1

    @staticmethod
    def avg_y(y):                                                  # models.py:609-628
        if y.ndim == 1:
            return np.copy(y)
        if y.ndim == 2:
            return np.average(y, axis=1)
        raise ValueError('y must be two-dimensional, not shape={}'.format(y.shape))

    @staticmethod
    def solve_sqrt(sqrt_mat, y, decomposition):
        """R^-1 y from a square root of R (models.py:460-479): the lower Cholesky factor, or for 'eig' either the
        tuple (eigenvalues, Q) or a square root S with R = S S^T."""
        from ._lib import DeviceMatrix
        if decomposition == 'cholesky':
            if isinstance(sqrt_mat, DeviceMatrix):
                return sqrt_mat._ctx.cho_solve(sqrt_mat, y)
            return cho_solve((sqrt_mat, True), y)
        if decomposition == 'eig':
            if isinstance(sqrt_mat, tuple):
                eig, Q = sqrt_mat
                proj = Q.T @ y
                return Q @ (proj / (eig if proj.ndim == 1 else eig[:, None]))
            return np.linalg.solve(sqrt_mat.T, np.linalg.solve(sqrt_mat, y))
        raise ValueError('decomposition must be either "cholesky" or "eig"')

    @staticmethod
    def compute_cov_factor(scale_sq, df):                          # models.py:490-503
        return cov_factor(scale_sq, df)

    @classmethod
    def _gram_pieces(cls, y, sqrt_R, basis, decomposition, dR=None):
        """G = [y | basis]^T R^-1 [y | basis] and, with dR (n x n x P), H_p = V^T dR_p V."""
        y2 = y[:, None] if y.ndim == 1 else y
        Z = np.concatenate([y2, basis], axis=1)
        V = cls.solve_sqrt(sqrt_R, Z, decomposition)
        H = None if dR is None else np.einsum('ia,ijp,jb->pab', V, dR, V)
        return Z.T @ V, H, y2.shape[1]

    @staticmethod
    def _need_dR(eval_gradient, dR):
        if eval_gradient and dR is None:
            raise ValueError('dR must be given if eval_gradient is True')

    @classmethod
    def compute_disp(cls, y, sqrt_R, basis, disp0, decomposition, eval_gradient=False, dR=None):   # models.py:234-278
        cls._need_dR(eval_gradient, dR)
        if np.all(disp0 == 0):
            if eval_gradient:
                return np.zeros_like(disp0), np.zeros((*disp0.shape, dR.shape[-1]))
            return np.zeros_like(disp0)
        G, H, ny = cls._gram_pieces(y, sqrt_R, basis, decomposition, dR if eval_gradient else None)
        disp = inv(inv(disp0) + ny * G[ny:, ny:])
        if eval_gradient:
            return disp, ny * np.einsum('ia,pab,bl->ilp', disp.T, H[:, ny:, ny:], disp)
        return disp

    @classmethod
    def compute_center(cls, y, sqrt_R, basis, center0, disp0, decomposition, eval_gradient=False, dR=None):   # :170-231
        cls._need_dR(eval_gradient, dR)
        if np.all(disp0 == 0):
            if eval_gradient:
                return np.copy(center0), np.zeros((*center0.shape, dR.shape[-1]))
            return np.copy(center0)
        G, H, ny = cls._gram_pieces(y, sqrt_R, basis, decomposition, dR if eval_gradient else None)
        w = np.full(ny, 1.0 / ny)
        disp = inv(inv(disp0) + ny * G[ny:, ny:])
        center = disp @ (np.linalg.solve(disp0, center0) + ny * (G[ny:, :ny] @ w))
        if eval_gradient:
            a_diff = np.concatenate([-w, center])                 # basis center - y_avg = [y | basis] a_diff
            return center, ny * disp @ np.einsum('pab,b->ap', H[:, ny:, :], a_diff)
        return center

    @classmethod
    def compute_df(cls, y, df0, eval_gradient=False, dR=None):     # models.py:281-307
        cls._need_dR(eval_gradient, dR)
        df = df0 + y.size
        if eval_gradient:
            return df, np.zeros(dR.shape[-1])
        return df

    @classmethod
    def compute_scale_sq(cls, y, sqrt_R, basis, center0, disp0, df0, scale0, decomposition,
                         eval_gradient=False, dR=None):            # models.py:387-457
        if df0 == np.inf:
            if eval_gradient:
                return scale0 ** 2, np.zeros(dR.shape[-1])
            return scale0 ** 2
        cls._need_dR(eval_gradient, dR)
        G, H, ny = cls._gram_pieces(y, sqrt_R, basis, decomposition, dR if eval_gradient else None)
        p = G.shape[0] - ny
        center0 = np.atleast_1d(np.asarray(center0, dtype=float))
        w = np.full(ny, 1.0 / ny)
        Gyy, Gby, Gbb = G[:ny, :ny], G[ny:, :ny], G[ny:, ny:]
        disp = np.zeros((p, p)) if np.all(disp0 == 0) else inv(inv(disp0) + ny * Gbb)
        quad = np.trace(Gyy) - ny * (w @ Gyy @ w)                  # tr((y - ybar)^T R^-1 (y - ybar))
        a_c = np.concatenate([w, -center0])                        # ybar - basis center0 = [y | basis] a_c
        s = G[ny:, :] @ a_c                                        # basis^T R^-1 (ybar - basis center0)
        quad2 = ny * (a_c @ G @ a_c - ny * (s @ disp @ s))         # Woodbury form of models.py:441-445, no n x n matrix
        df = df0 + y.size
        scale_sq = (df0 * scale0 ** 2 + quad + quad2) / df
        if eval_gradient:
            Cc = np.concatenate([np.eye(ny) - np.outer(w, np.ones(ny)), np.zeros((p, ny))], axis=0)
            a_m = ny * (a_c - ny * np.concatenate([np.zeros(ny), disp @ s]))
            d = np.array([-np.trace(Cc.T @ Hp @ Cc) - (a_m @ Hp @ a_m) / ny for Hp in H]) / df
            return scale_sq, d
        return scale_sq

    # -- priors (models.py:153-167) ------------------------------------------------------------
    @property
    def center0(self):
        return self._center_0

    @property
    def disp0(self):
        return self._disp_0

    @property
    def df0(self):
        return self._df_0

    @property
    def scale0(self):
        return self._scale_0

    # -- device plumbing -----------------------------------------------------------------------
    def _context(self):
        if self._ctx is None:
            if self.backend == "cpu":
                from ._cpu import cpu_context
                self._ctx = cpu_context()
            else:
                self._ctx = default_context(self.device)
        return self._ctx

    def _check_decomposition(self):
        if self.decomposition not in ('cholesky', 'eig'):
            raise ValueError('decomposition must be "cholesky" or "eig"')     # models.py:719, 976

    @staticmethod
    def _rhs(X, y):
        """[Y | B]: the curves and the constant basis column, the fused kernel's right-hand sides."""
        y = np.asarray(y, dtype=float)
        if y.ndim == 1:
            y = y[:, None]
        if y.ndim != 2:
            raise ValueError('y must be two-dimensional, not shape={}'.format(y.shape))
        if y.shape[1] + 1 > GSUM_MAX_RHS:
            raise ValueError(f"at most {GSUM_MAX_RHS - 1} curves are supported, got {y.shape[1]}")
        return np.concatenate([y, np.ones((y.shape[0], 1))], axis=1)

    # -- more curves than one device call takes (GSUM_MAX_RHS - 1 = 15) ---------------------------------------------------------------------
    # The reference takes any number of curves (models.py:602-628, 1026-1035).  Everything its algebra does with the Gram matrix G = Z^T R^-1 Z,
    # Z = [Y | 1] -- and with the gradient pieces H_p = V^T dR_p V -- uses the curve block only through its TRACE, the SUM of its entries and its
    # cross terms with the basis column (posterior_from_gram, lml_grad_from_gram, hyper_gradients_from_gram: uniform weights over the curves).  So
    # the curves go to the device in chunks [<= 13 curves | the sum of ALL curves | 1]: the chunks' diagonals are the curve block's diagonal, the
    # sum column gives the sum of its entries, and a matrix with those entries on the diagonal, the right total and one constant everywhere else
    # stands in for it.  Single evaluations (fit, likelihood, gradient, predict) and the truncation classes' surfaces (truncation.py: _grid_chunked).
    @staticmethod
    def _many_curves(y):
        y = np.asarray(y)
        return y.ndim == 2 and y.shape[1] + 1 > GSUM_MAX_RHS

    @staticmethod
    def _rhs_chunks(X, y):
        y = np.asarray(y, dtype=float)
        n, r = y.shape
        tail = np.concatenate([y.sum(axis=1)[:, None], np.ones((n, 1))], axis=1)
        step = GSUM_MAX_RHS - 3
        for lo in range(0, r, step):
            idx = np.arange(lo, min(lo + step, r))
            yield idx, np.concatenate([y[:, idx], tail], axis=1)

    @staticmethod
    def _stand_in(blocks, ny):
        """The (ny + 1) x (ny + 1) stand-in for Z^T M Z from the chunks' (m + 2) x (m + 2) matrices of [curves | sum | 1]."""
        out = np.zeros((ny + 1, ny + 1))
        total = corner = 0.0
        for idx, M in blocks:
            m = len(idx)
            out[idx, idx] = np.diag(M)[:m]
            out[idx, ny] = out[ny, idx] = M[:m, m + 1]
            total, corner = M[m, m], M[m + 1, m + 1]              # (the same in every chunk)
        off = (total - np.trace(out[:ny, :ny])) / (ny * (ny - 1))
        out[:ny, :ny] += off * (1.0 - np.eye(ny))
        out[ny, ny] = corner
        return out

    def _active_kernel(self):
        if getattr(self, 'kernel_', None) is not None:
            return self.kernel_
        return self._default_kernel if self.kernel is None else self.kernel      # models.py:946-952

    # -- lazily materialised n x n attributes (device -> host only on access) -------------------
    @property
    def corr_(self):
        if self._corr is None and self._fit:
            X = np.asarray(self.X_train_, dtype=float)
            self._corr = self._context().kernel_matrix(describe_kernel(self.kernel_, X.shape[1]), X)
        return self._corr

    @property
    def _eigh_tuple_(self):
        """(eig, Q) of corr_ + nugget I (models.py:714-715) -- an attribute of the 'eig' mode only, computed on the host when it is read."""
        if self.decomposition != 'eig' or not self._fit:
            return None
        if getattr(self, '_eigh_cache', None) is None:
            from scipy.linalg import eigh
            C = np.array(self.corr_, dtype=float)
            self._eigh_cache = eigh(C + self.nugget * np.eye(C.shape[0]))
        return self._eigh_cache

    @property
    def corr_L_(self):
        if self.decomposition == 'eig':                      # models.py:717: Q @ diag(sqrt(eig)), not triangular
            tup = self._eigh_tuple_
            return None if tup is None else tup[1] @ np.diag(np.sqrt(tup[0]))
        if self._corr_L is None and self._L_dev is not None:
            self._corr_L = self._L_dev.to_host()
        return self._corr_L

    corr_sqrt_ = corr_L_

    # -- log marginal likelihood (models.py:912-1039) --------------------------------------------
    def log_marginal_likelihood(self, theta=None, eval_gradient=False, X=None, y=None):
        if theta is None and self._fit:
            if eval_gradient:
                raise ValueError("Gradient can only be evaluated for theta!=None")   # models.py:940-943
            return self.log_marginal_likelihood_value_
        self._check_decomposition()
        kernel = self._active_kernel()
        X = self.X_train_ if X is None else X
        y = self.y_train_ if y is None else y
        X = np.asarray(X, dtype=float)
        if theta is not None:
            # models.py:953 without scikit-learn's clone (0.1-0.4 ms of get_params / set_params per call -- more than the device takes at the
            # reference's own sizes): same descriptor and gradient parameters, byte for byte (tests/test_host_logic.py)
            desc = describe_thetas(kernel, [theta], X.shape[1])[0]
        else:
            desc = describe_kernel(kernel, X.shape[1])
        if self._many_curves(y):
            return self._lml_many_curves(desc, kernel, theta, eval_gradient, X, np.asarray(y, dtype=float))
        Z = self._rhs(X, y)
        if eval_gradient:                                                            # models.py:957-958, 1041-1056
            params = describe_gradients(kernel, [theta], X.shape[1])[0] if theta is not None else describe_gradient(kernel, X.shape[1])
            if not params:
                return self.log_marginal_likelihood(theta, X=X, y=y), np.zeros(0)
            G, sld, info, trace, H = self._context().lml_grad(desc, params, X, Z, self.nugget)
            if info != 0:
                return -np.inf, np.zeros(len(params))                                # models.py:970-972
            return self._lml_grad_gram(G, sld, trace, H, X.shape[0])
        G, sld, info = self._context().lml_batch([desc], X, Z, self.nugget)
        if info[0] != 0:
            return -np.inf                                                           # models.py:970-972
        lml, _ = self._lml_gram(G[0], sld[0], X.shape[0])
        return lml

    def _lml_many_curves(self, desc, kernel, theta, eval_gradient, X, y):
        """log_marginal_likelihood for more than GSUM_MAX_RHS - 1 curves: one device call per chunk of curves (see _rhs_chunks)."""
        ny, ctx = y.shape[1], self._context()
        params = None
        if eval_gradient:
            params = describe_gradients(kernel, [theta], X.shape[1])[0] if theta is not None else describe_gradient(kernel, X.shape[1])
            if not params:
                return self.log_marginal_likelihood(theta, X=X, y=y), np.zeros(0)
        Gb, Hb, sld, trace = [], [], None, None
        for idx, Zc in self._rhs_chunks(X, y):
            if eval_gradient:
                Gc, sld, info, trace, Hc = ctx.lml_grad(desc, params, X, Zc, self.nugget)
                Hb.append((idx, Hc))
            else:
                Gc, slds, infos = ctx.lml_batch([desc], X, Zc, self.nugget)
                Gc, sld, info = Gc[0], slds[0], infos[0]
            if info != 0:
                return (-np.inf, np.zeros(len(params))) if eval_gradient else -np.inf     # models.py:970-972
            Gb.append((idx, Gc))
        G = self._stand_in(Gb, ny)
        if not eval_gradient:
            return self._lml_gram(G, sld, X.shape[0])[0]
        H = np.array([self._stand_in([(idx, Hc[p]) for idx, Hc in Hb], ny) for p in range(len(params))])
        return self._lml_grad_gram(G, sld, trace, H, X.shape[0])

    def _lml_gram(self, G, sld, n_points):
        """(log marginal likelihood, posterior dict) of this process from one Gram matrix."""
        return lml_from_gram(G, sld, n_points, self.center0, self.disp0, self.df0, self.scale0)

    def _lml_gram_batch(self, G, sld, n_points):
        return lml_from_gram_batch(G, sld, n_points, self.center0, self.disp0, self.df0, self.scale0)

    def _lml_gram_batch_sd(self, G, sld, n_points, sds):
        """The same with the prior replaced by ``sd=sds[b]`` per entry (df0 = inf, scale0 = sd: models.py:115-117)."""
        return lml_from_gram_batch(G, sld, n_points, self.center0, self.disp0, np.inf, np.asarray(sds, dtype=float))

    def _lml_grad_gram(self, G, sld, trace, H, n_points):
        return lml_grad_from_gram(G, sld, trace, H, n_points, self.center0, self.disp0, self.df0, self.scale0)

    def _cov_terms(self, d):
        """(factor, descriptor) such that the two-argument covariance of the fitted process is
        factor * kernel_desc(X, Xp): what TruncationProcess conditions with (models.py:599, 1343)."""
        return self.cov_factor_, describe_kernel(self.kernel_, d).without_white()      # kernel_(X, Xp), both given: no white noise

    # -- fit (models.py:630-738) -------------------------------------------------------------------
    def _constrained_optimization(self, obj_func, initial_theta, bounds, warn=None):
        """models.py:884-900.  ``warn``: where the convergence message goes instead of ``warnings.warn`` (the lock-step restarts
        collect theirs and emit them from the calling thread: ``warnings.catch_warnings`` is process-global state)."""
        if self.optimizer == "fmin_l_bfgs_b":
            theta_opt, func_min, info = fmin_l_bfgs_b(obj_func, initial_theta, bounds=bounds)
            if info["warnflag"] != 0:
                (warn or warnings.warn)("fmin_l_bfgs_b terminated abnormally with the  state: %s" % info, ConvergenceWarning)
        elif callable(self.optimizer):
            theta_opt, func_min = self.optimizer(obj_func, initial_theta, bounds=bounds)
        else:
            raise ValueError("Unknown optimizer %s." % self.optimizer)
        return theta_opt, func_min

    def log_marginal_likelihood_batch(self, thetas, X=None, y=None):
        """``log_marginal_likelihood(theta, eval_gradient=True)`` for several ``theta`` at once: [(lml, grad), ...].  The
        evaluations are independent and go to the device as one pipelined batch (gsum_lml_grad_batch); every entry equals
        the single call's result bit for bit."""
        self._check_decomposition()
        base = self._active_kernel()
        X = np.asarray(self.X_train_ if X is None else X, dtype=float)
        y = self.y_train_ if y is None else y
        if self._many_curves(y):
            return [self.log_marginal_likelihood(t, eval_gradient=True, X=X, y=y) for t in thetas]
        Z = self._rhs(X, y)
        thetas = [np.atleast_1d(np.asarray(t, dtype=float)) for t in thetas]
        # descriptors and gradient parameters straight from theta (kernels.describe_thetas): no scikit-learn clone per start
        params = describe_gradients(base, thetas, X.shape[1])              # the weights carry hyperparameter values: per theta
        if not thetas or not params[0] or len(thetas) == 1:
            return [self.log_marginal_likelihood(t, eval_gradient=True, X=X, y=y) for t in thetas]
        descs = describe_thetas(base, thetas, X.shape[1])
        G, sld, info, trace, H = self._context().lml_grad_batch(descs, params, X, Z, self.nugget)
        out = []
        for i, t in enumerate(thetas):
            if info[i] != 0:
                out.append((-np.inf, np.zeros_like(t)))                              # models.py:970-972
            else:
                out.append(self._lml_grad_gram(G[i], sld[i], trace[i], H[i], X.shape[0]))
        return out

    def _lockstep_restarts(self, starts, bounds):
        """All starts of a multi-start fit (models.py:641-662) advanced together: every start runs its own L-BFGS in a thread,
        the objective calls of one sweep meet in ``log_marginal_likelihood_batch`` and are evaluated on the device as ONE
        pipelined batch.  Each start sees exactly the values the sequential loop would give it, so the optima are the same."""
        import threading
        n = len(starts)
        cond = threading.Condition()
        pending, results, alive = {}, {}, set(range(n))
        state = {"error": None}

        def flush():                       # called with the lock held: every live start has asked
            ids = sorted(pending)
            try:
                vals = self.log_marginal_likelihood_batch([pending[i] for i in ids])
            except BaseException as exc:   # noqa: BLE001 -- handed to every waiting thread
                state["error"] = exc
                vals = [(np.nan, None)] * len(ids)
            for i, v in zip(ids, vals):
                results[i] = v
            pending.clear()
            cond.notify_all()

        def make_obj(i):
            def obj(theta, eval_gradient=True):
                with cond:
                    pending[i] = np.array(theta, dtype=float)
                    if len(pending) == len(alive):
                        flush()
                    else:
                        while i not in results and state["error"] is None:
                            cond.wait()
                    if state["error"] is not None:
                        raise state["error"]
                    lml, grad = results.pop(i)
                return -lml, -grad
            return obj

        optima = [None] * n

        caught = [[] for _ in range(n)]       # convergence messages of a start: collected in its thread, emitted by the caller's

        def run(i):
            try:
                optima[i] = self._constrained_optimization(make_obj(i), starts[i], bounds,
                                                           warn=lambda msg, cat, i=i: caught[i].append((msg, cat)))
            except BaseException as exc:   # noqa: BLE001
                with cond:
                    if state["error"] is None:
                        state["error"] = exc
                    cond.notify_all()
            finally:
                with cond:
                    alive.discard(i)
                    if pending and len(pending) == len(alive):
                        flush()

        threads = [threading.Thread(target=run, args=(i,), daemon=True) for i in range(n)]
        for t in threads:
            t.start()
        try:
            for t in threads:
                t.join()
        except BaseException as exc:       # KeyboardInterrupt in the caller's thread: wake and end the parked workers, then re-raise
            with cond:
                if state["error"] is None:
                    state["error"] = exc
                cond.notify_all()
            for t in threads:
                t.join(timeout=5.0)
            raise
        if state["error"] is not None:
            raise state["error"]
        for rec in caught:                 # in start order, from the calling thread (no catch_warnings in the workers: its
            for msg, cat in rec:           # save / restore of warnings.filters is process-global and the starts end out of order)
                warnings.warn(msg, cat)
        return optima

    def _calibrate_kernel(self):
        """models.py:630-669, with the intended argmin over restarts (the reference's ragged
        ``np.array(optima)`` at :664 raises on numpy >= 1.24).  With ``n_restarts_optimizer > 0`` and the default optimiser all
        starts advance in lock step, their objective evaluations batched on the device (``_lockstep_restarts``); the random
        initial points are drawn in the reference's order, so the starts -- and the optima -- are the sequential loop's."""
        if self.optimizer is not None and self.kernel_.n_dims > 0:
            def obj_func(theta, eval_gradient=True):                                   # models.py:634-640
                if eval_gradient:
                    lml, grad = self.log_marginal_likelihood(theta, eval_gradient=True)
                    return -lml, -grad
                return -self.log_marginal_likelihood(theta)

            starts = [np.array(self.kernel_.theta, dtype=float)]
            bounds = self.kernel_.bounds
            if self.n_restarts_optimizer > 0:
                if not np.isfinite(self.kernel_.bounds).all():
                    raise ValueError("Multiple optimizer restarts (n_restarts_optimizer>0) "
                                     "requires that all bounds are finite.")
                for _ in range(self.n_restarts_optimizer):
                    starts.append(self._rng.uniform(bounds[:, 0], bounds[:, 1]))
            if len(starts) > 1 and self.optimizer == "fmin_l_bfgs_b" and self.batch_restarts:
                optima = self._lockstep_restarts(starts, bounds)
            else:
                optima = [self._constrained_optimization(obj_func, st, bounds) for st in starts]
            values = [o[1] for o in optima]
            best = int(np.argmin(values))
            self.kernel_.theta = optima[best][0]
            return -float(values[best])
        return None

    def fit(self, X, y):
        self._check_decomposition()
        self.kernel_ = clone(self._default_kernel if self.kernel is None else self.kernel)   # models.py:685-688
        self._rng = check_random_state(self.random_state)
        if self.copy_X_train:
            self.X_train_ = np.copy(X)
            self.y_train_ = np.copy(y)
        else:
            self.X_train_, self.y_train_ = X, y
        self.basis_train_ = self.basis(self.X_train_)
        self._fit = False
        self._corr = self._corr_L = self._eigh_cache = None
        Xd = np.asarray(self.X_train_, dtype=float)
        many = self._many_curves(self.y_train_)
        Z = None if many else self._rhs(Xd, self.y_train_)

        lml_opt = self._calibrate_kernel()                                  # models.py:707

        # one build + one factorisation serve both the likelihood value and the posterior updates
        # (the reference does each twice, models.py:668-669 and :708-711)
        ctx = self._context()
        desc = describe_kernel(self.kernel_, Xd.shape[1])
        if self._L_dev is not None:
            self._L_dev.free()
        for rep in self._replicas.values():
            rep.free()
        self._replicas = {}
        self._L_dev, info = ctx.factorize(desc, Xd, diag_add=self.nugget)
        if info != 0:
            self._L_dev.free()
            self._L_dev = None
            raise np.linalg.LinAlgError("Matrix is not positive definite")    # numpy's message; models.py:711
        if many:                                  # one forward solve per chunk of curves on the one factor (see _rhs_chunks)
            y2 = np.asarray(self.y_train_, dtype=float)
            blocks, sld = [], None
            for idx, Zc in self._rhs_chunks(Xd, y2):
                Gc, sld = ctx.forward_gram(self._L_dev, Zc)
                blocks.append((idx, Gc))
            G = self._stand_in(blocks, y2.shape[1])
        else:
            G, sld = ctx.forward_gram(self._L_dev, Z)
        lml, post = self._lml_gram(G, sld, Xd.shape[0])
        self.log_marginal_likelihood_value_ = lml if lml_opt is None else lml_opt
        self._gram = (G, sld)
        self.center_ = post["center"]                                        # models.py:721-736
        self.disp_ = post["disp"]
        self.df_ = post["df"]
        self.scale_ = np.sqrt(post["scale_sq"])
        self.cov_factor_ = self.cbar_sq_mean_ = post["cov_factor"]
        self._fit = True
        return self

    # -- accessors that the reference recomputes from the factor (models.py:505-549) --------------
    def _posterior(self):
        G, _ = self._gram
        return posterior_from_gram(G, np.shape(self.X_train_)[0], self.center0, self.disp0, self.df0, self.scale0)

    def center(self):
        self._check_decomposition()
        return self._posterior()["center"]

    def disp(self):
        self._check_decomposition()
        return self._posterior()["disp"]

    def df(self):
        return self.df0 + np.asarray(self.y_train_).size

    def scale(self):
        self._check_decomposition()
        return np.sqrt(self._posterior()["scale_sq"])

    # -- mean / cov (models.py:551-599) ---------------------------------------------------------------
    def mean(self, X):
        center = self.center_ if self._fit else self.center0
        return self.basis(X) @ center

    def _cov_parts(self, d):
        """(factor, descriptor): the process covariance is ``factor * kernel_desc(X[, Xp])`` -- prior quantities before ``fit``
        (models.py:579-589), ``cov_factor_`` and the fitted kernel after (:590-599).  One-argument form: the descriptor's own white
        noise stays in."""
        if not self._fit:
            if self.df0 <= 2:
                raise ValueError('df must be greater than 2 for the covariance to exist')
            return cov_factor(self.scale0 ** 2, self.df0), describe_kernel(self._default_kernel if self.kernel is None else self.kernel, d)
        return self.cov_factor_, describe_kernel(self.kernel_, d)

    def cov(self, X, Xp=None):
        X = np.asarray(X, dtype=float)
        factor, desc = self._cov_parts(X.shape[1])
        return factor * self._context().kernel_matrix(desc, X, None if Xp is None else np.asarray(Xp, dtype=float))

    def underlying_properties(self, X, return_std=False, return_cov=False):
        y_mean = self.mean(X)
        if return_cov:
            return y_mean, self.cov(X)
        if return_std:                               # diag of the one-argument kernel: one number (leaves exactly 1, WhiteKernel noise in)
            X = np.asarray(X, dtype=float)
            factor, desc = self._cov_parts(X.shape[1])
            return y_mean, np.sqrt(np.full(X.shape[0], factor * desc.one_arg_diagonal(X)))      # (a DotProduct leaf: one value per point)
        return y_mean

    # -- predict (models.py:753-845) ------------------------------------------------
    def predict(self, X, return_std=False, return_cov=False, Xc=None, y=None, pred_noise=False, devices=None):
        """``devices`` (additive; ``"all"`` or a list of GPU indices): the new points are cut into one block per device
        (``gsum_shard_range``), every device factorises its own copy of the training matrix once (kept until the next ``fit``) and
        the blocks run side by side, one host thread per device -- columns of the predictive covariance are independent per new
        point (models.py:836), so mean and standard deviation equal the one-device call; ``return_cov`` needs all columns on one
        device and is refused."""
        return self._predict_core(X, return_std, return_cov, Xc, y, pred_noise, False, devices=devices)[0]

    def _group(self, devices):
        """The device group of ``devices=`` (process-wide, adopting the default contexts; ``backend='cpu'``: as many CPU contexts)."""
        if self.backend == "cpu":
            from ._cpu import cpu_group          # no devices to count: "all" means two members, a list as many as it names
            return cpu_group(2 if isinstance(devices, str) else (1 if np.isscalar(devices) else len(devices)))
        from ._lib import default_group
        return default_group(devices)

    def _replica_factor(self, ctx, desc, Xc):
        """The factor of the training matrix on another device's context (built there once per fit)."""
        if ctx is self._context():
            return self._L_dev
        key = id(ctx)
        L = self._replicas.get(key)
        if L is None or getattr(L, "_h", True) is None:
            L, info = ctx.factorize(desc, Xc, diag_add=self.nugget)
            if info != 0:
                L.free()
                raise np.linalg.LinAlgError("Matrix is not positive definite")
            self._replicas[key] = L
        return L

    def _predict_core(self, X, return_std, return_cov, Xc, y, pred_noise, want_basis, devices=None):
        """predict, plus (want_basis) the conditional basis 1 - R_no R^-1 1 of models.py:1168 from the same
        triangular solve: one more right-hand-side column."""
        if return_std and return_cov:
            raise RuntimeError('Only one of return_std or return_cov may be True')
        if not self._fit:
            return self.underlying_properties(X=X, return_std=return_std, return_cov=return_cov), None
        self._check_decomposition()
        if devices is not None and return_cov:
            raise ValueError("return_cov needs every new point on one device: call predict without devices=")
        ctx = self._context()
        X = np.asarray(X, dtype=float)
        desc = describe_kernel(self.kernel_, X.shape[1])
        own = None
        fitted_inputs = Xc is None
        if Xc is None:
            Xc = np.asarray(self.X_train_, dtype=float)
            L = self._L_dev
        elif devices is not None:
            Xc = np.asarray(Xc, dtype=float)
            L = None
        else:
            Xc = np.asarray(Xc, dtype=float)
            own, info = ctx.factorize(desc, Xc, diag_add=self.nugget)             # models.py:807
            L = own
            if info != 0:
                L.free()
                raise np.linalg.LinAlgError("Matrix is not positive definite")    # models.py:809
        try:
            y = self.y_train_ if y is None else y
            y = np.asarray(y, dtype=float)
            if y.ndim == 1:
                y = y[:, None]
            m_old = self.mean(Xc)                                                 # models.py:818
            m_new = self.mean(X)                                                  # models.py:819
            resid = y - m_old[:, None]
            n_curves = resid.shape[1]
            if want_basis:
                resid = np.concatenate([resid, self.basis(Xc)], axis=1)

            def terms(ctx_, L_, Xs, want_cov_):
                parts, css, cv_ = [], None, None
                for lo in range(0, resid.shape[1], GSUM_MAX_RHS):
                    wc = want_cov_ and lo == 0
                    css, VtW, cv = ctx_.predict_terms(L_, desc, Xc, Xs, rhs=resid[:, lo:lo + GSUM_MAX_RHS], want_cov=wc)
                    parts.append(VtW)
                    cv_ = cv if wc else cv_
                return np.concatenate(parts, axis=1), css, cv_

            if devices is None:
                shifts, colsumsq, VtV = terms(ctx, L, X, return_cov)
            else:
                from .grid import shard_range
                grp = self._group(devices)
                world = len(grp)

                def block(r, ctx_r):
                    lo, hi = shard_range(X.shape[0], r, world)
                    if hi == lo:
                        return np.empty((0, resid.shape[1])), np.empty(0)
                    if fitted_inputs:
                        L_r, mine = self._replica_factor(ctx_r, desc, Xc), None
                    else:
                        L_r, info = ctx_r.factorize(desc, Xc, diag_add=self.nugget)          # models.py:807, on this block's device
                        mine = L_r
                        if info != 0:
                            L_r.free()
                            raise np.linalg.LinAlgError("Matrix is not positive definite")
                    try:
                        return terms(ctx_r, L_r, X[lo:hi], False)[:2]
                    finally:
                        if mine is not None:
                            mine.free()
                blocks = grp.map(block)
                shifts = np.concatenate([b[0] for b in blocks], axis=0)
                colsumsq = np.concatenate([b[1] for b in blocks])
                VtV = None
        finally:
            if own is not None:
                own.free()
        # R_no R^-1 (y - m) = (L^-1 R_on)^T (L^-1 (y - m))                         models.py:831-832
        cond_basis = self.basis(X) - shifts[:, n_curves:] if want_basis else None
        m_pred = np.squeeze(m_new[:, None] + shifts[:, :n_curves])
        if return_std or return_cov:
            var = cov_factor(self.scale_ ** 2, self.df_)                           # models.py:840
            if return_std:
                # diag of the one-argument kernel: unit base value, WhiteKernel noise included (:824)
                diag_nn = desc.one_arg_diagonal(X)
                r_diag = diag_nn - colsumsq                                        # models.py:836
                if pred_noise:
                    r_diag = r_diag + self.nugget                                  # models.py:837-838
                return (m_pred, np.sqrt(np.squeeze(var * r_diag))), cond_basis    # models.py:841-843
            R_pred = ctx.kernel_matrix(desc, X) - VtV
            if pred_noise:
                R_pred += self.nugget * np.eye(len(X))
            return (m_pred, np.squeeze(var * R_pred)), cond_basis
        return m_pred, cond_basis

    # -- sampling (models.py:847-879) ----------------------------------------------------------------------
    def sample_y(self, X, n_samples=1, random_state=0, underlying=False, method='svd', jitter=0.0):
        """``method='svd'`` is the reference's ``rng.multivariate_normal`` on the host (an n x n SVD);
        ``method='cholesky'`` factorises ``cov + jitter I`` on the device and returns ``mean + L z`` (additive
        option: the same distribution, not the same numbers for a given seed)."""
        rng = check_random_state(random_state)
        if underlying:
            y_mean, y_cov = self.underlying_properties(X=X, return_cov=True)
        else:
            y_mean, y_cov = self.predict(X, return_cov=True)
        if method == 'cholesky':
            ctx = self._context()
            m = y_cov.shape[0]
            L = ctx.upload(y_cov + jitter * np.eye(m))
            try:
                if ctx.potrf(L) != 0:
                    raise np.linalg.LinAlgError("predictive covariance is not positive definite: pass jitter > 0")
                cols = 1 if y_mean.ndim == 1 else y_mean.shape[1]
                draws = ctx.tri_multiply(L, rng.standard_normal((m, n_samples * cols)))
            finally:
                L.free()
            if y_mean.ndim == 1:
                return y_mean[:, None] + draws
            return np.hstack([(y_mean[:, i:i + 1] + draws[:, i * n_samples:(i + 1) * n_samples])[:, np.newaxis]
                              for i in range(cols)])
        if method != 'svd':
            raise ValueError('method must be "svd" or "cholesky"')
        if y_mean.ndim == 1:
            return rng.multivariate_normal(y_mean, y_cov, n_samples).T
        samples = [rng.multivariate_normal(y_mean[:, i], y_cov, n_samples).T[:, np.newaxis]
                   for i in range(y_mean.shape[1])]
        return np.hstack(samples)


class ConjugateStudentProcess(ConjugateGaussianProcess):
    """Conjugate-prior Student-t process; same surface as gsum.ConjugateStudentProcess (models.py:1091-1273).

    It shares fit and the device path with the Gaussian class (the reference's BaseConjugateProcess); what differs
    is host algebra on the same Gram matrix: the likelihood is a ratio of normalisation constants, and covariances
    carry the rank-one term basis disp basis^T from integrating out the mean."""

    def _lml_gram(self, G, sld, n_points):
        return student_lml_from_gram(G, sld, n_points, self.center0, self.disp0, self.df0, self.scale0)

    def _lml_gram_batch(self, G, sld, n_points):
        return np.array([self._lml_gram(Gi, si, n_points)[0] for Gi, si in zip(G, sld)])

    def _lml_gram_batch_sd(self, G, sld, n_points, sds):
        # an infinite prior df has no Student-t normalisation constant (log_norm diverges, models.py:1234-1240); the
        # reference would return nan there as well
        return np.array([student_lml_from_gram(Gi, si, n_points, self.center0, self.disp0, np.inf, sd)[0]
                         for Gi, si, sd in zip(G, sld, sds)])

    def _lml_grad_gram(self, G, sld, trace, H, n_points):
        return student_lml_grad_from_gram(G, sld, trace, H, n_points, self.center0, self.disp0, self.df0, self.scale0)

    def _cov_terms(self, d):
        # var * (corr + basis disp basis^T) with a constant basis is an additive constant in the kernel (:1125)
        _, desc = super()._cov_terms(d)
        return cov_factor(self.scale_ ** 2, self.df_), desc.plus_constant(float(self.disp_[0, 0]))

    def _cov_parts(self, d):                                             # models.py:1099-1125
        if not self._fit:
            df, scale, disp = self.df0, self.scale0, self.disp0
            kernel = self._default_kernel if self.kernel is None else self.kernel
        else:
            df, scale, disp = self.df_, self.scale_, self.disp_
            kernel = self.kernel_
        if df <= 2:
            raise ValueError('df must be greater than 2 for the covariance to exist')
        # corr + basis disp basis^T with the constant basis: an additive constant on top of the kernel
        return cov_factor(scale ** 2, df), describe_kernel(kernel, d).plus_constant(float(np.atleast_2d(disp)[0, 0]))

    def predict(self, X, return_std=False, return_cov=False, Xc=None, y=None, pred_noise=False):   # models.py:1127-1184
        pred, basis = self._predict_core(X, return_std, return_cov, Xc, y, pred_noise, True)
        if not self._fit:
            disp = self.disp0
            var = cov_factor(self.scale0 ** 2, self.df0)
            basis = self.basis(X)
        else:
            disp = self.disp_
            var = self.cov_factor_
        mean_cov = var * (basis @ disp @ basis.T)                         # from integrating out the mean, :1174
        if return_std:
            mean, std = pred
            return mean, std + np.sqrt(np.diag(mean_cov))                # standard deviations are added (:1177)
        if return_cov:
            mean, cov = pred
            return mean, cov + mean_cov
        return pred
