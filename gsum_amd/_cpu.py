"""``backend='cpu'``: the same operator interface as :class:`gsum_amd._lib.HipContext`, on numpy / scipy / scikit-learn.

This is BASELINE config 1 ("scipy CPU path, plumbing, no GPU") and SURVEY.md section 8(b)'s additive ``backend=`` switch:
the reference's own third-party routines behind the interface the HIP library replaces --

    kernel(X[, Y])                    sklearn RBF / Matern leaves       gsum/models.py:708, 822-824, 958-960
    numpy.linalg.cholesky             LAPACK dpotrf (with its info)     gsum/models.py:711, 809, 969
    scipy.linalg.cho_solve halves     dtrtrs                            gsum/models.py:479, 831, 836, 1032

It is never chosen silently: the default backend is ``'hip'`` and a missing library or GPU raises there.  Select it with
``ConjugateGaussianProcess(..., backend='cpu')`` / ``TruncationGP(..., backend='cpu')`` or ``GSUM_BACKEND=cpu``.  It does not
import ``oracle/`` (test infrastructure) and the GPU tests never run through it.  Every method counts its calls in
``self.calls`` (the multi-process CPU tests read how many factorisations a rank ran).
"""
from __future__ import annotations

import collections

import numpy as np
from scipy.linalg import cho_solve as _cho_solve
from scipy.linalg import solve_triangular
from scipy.linalg.lapack import dpotrf
from sklearn.gaussian_process.kernels import RBF, ConstantKernel, DotProduct, ExpSineSquared, Matern, RationalQuadratic

from ._lib import FAMILY, GSUM_MAX_RHS, OP_ADD, OP_CONST, OP_LEAF, OP_POW, OP_WHITE, GradParam, KernelDesc
from .series import geometric_sum

_NU = {FAMILY["rbf"]: None, FAMILY["matern52"]: 2.5, FAMILY["matern32"]: 1.5, FAMILY["matern12"]: 0.5, FAMILY["matern_inf"]: np.inf}


def _leaf(desc, d: int, free=False):
    """The stationary scikit-learn leaf of a flattened descriptor / of one leaf of a tree (length scale(s) as it holds them)."""
    ls = np.array(desc.length_scale[:d]) if desc.anisotropic else float(desc.length_scale[0])
    bounds = (1e-300, 1e300) if free else "fixed"
    if int(desc.family) == FAMILY["rq"]:
        return RationalQuadratic(length_scale=ls, alpha=float(desc.alpha), length_scale_bounds=bounds, alpha_bounds=bounds)
    if int(desc.family) == FAMILY["dot"]:
        return DotProduct(sigma_0=float(desc.length_scale[0]), sigma_0_bounds=bounds)
    if int(desc.family) == FAMILY["expsine"]:
        return ExpSineSquared(length_scale=ls, periodicity=float(desc.alpha), length_scale_bounds=bounds, periodicity_bounds=bounds)
    nu = _NU[int(desc.family)]
    return RBF(ls, length_scale_bounds=bounds) if nu is None else Matern(ls, length_scale_bounds=bounds, nu=nu)


def _tree_matrix(desc, X, Y, param=None):
    """The postfix program of a tree descriptor on arrays, scikit-learn evaluating the leaves: the matrix, and with ``param`` (a
    GradParam of the TREE_* kind) its derivative with respect to that log-hyperparameter (sum and product rules)."""
    X = np.asarray(X, dtype=float)
    Yv = None if Y is None else np.asarray(Y, dtype=float)
    shape = (X.shape[0], X.shape[0] if Yv is None else Yv.shape[0])
    eye = np.eye(shape[0]) if Yv is None else np.zeros(shape)
    zero = np.zeros(shape)
    stack = []
    for k in range(desc.n_ops):
        op = desc.op[k]
        if op >= OP_POW:                       # Exponentiation (kernels.py: K ** exponent, K_gradient *= exponent K ** (exponent - 1))
            e = desc.cval[op - OP_POW]
            a, da = stack.pop()
            stack.append((a ** e, da * (e * a ** (e - 1))))
        elif op >= OP_WHITE:
            c = op - OP_WHITE
            hit = param is not None and param.code == GradParam.TREE_WHITE and param.dim == c
            stack.append((desc.cval[c] * eye, desc.cval[c] * eye if hit else zero))
        elif op >= OP_CONST:
            c = op - OP_CONST
            hit = param is not None and param.code == GradParam.TREE_CONST and param.dim == c
            stack.append((np.full(shape, desc.cval[c]), np.full(shape, desc.cval[c]) if hit else zero))
        elif op >= OP_LEAF:
            l = op - OP_LEAF
            mine = param is not None and param.code >= GradParam.TREE_LENGTH_ISO and (param.dim >> 4) == l
            if mine and Yv is None:
                V, dK = _leaf(desc.leaf[l], X.shape[1], free=True)(X, eval_gradient=True)
                # scikit-learn's theta order within a leaf: alphabetical -- RationalQuadratic: alpha, length_scale; else length_scale[s]
                if int(desc.leaf[l].family) == FAMILY["rq"]:
                    dv = dK[:, :, 0] if param.code == GradParam.TREE_ALPHA else dK[:, :, 1]
                elif int(desc.leaf[l].family) == FAMILY["expsine"]:         # length_scale, periodicity
                    dv = dK[:, :, 1] if param.code == GradParam.TREE_ALPHA else dK[:, :, 0]
                else:
                    dv = dK[:, :, param.dim & 15] if param.code == GradParam.TREE_LENGTH_DIM else dK[:, :, 0]
                stack.append((V, dv))
            else:
                stack.append((_leaf(desc.leaf[l], X.shape[1])(X, Yv), zero))
        else:
            (b, db), (a, da) = stack.pop(), stack.pop()
            stack.append((a + b, da + db) if op == OP_ADD else (a * b, da * b + a * db))
    return stack[0]


def kernel_matrix(desc: KernelDesc, X, Y=None, diag_add=0.0):
    """amplitude * leaf(X[, Y]) (+ white noise on the one-argument diagonal) + additive constant (+ diag_add): the kernel-build
    kernel's arithmetic, entry for entry (csrc/kernels/build.hip.h, k_build2), with scikit-learn evaluating the leaf."""
    X = np.asarray(X, dtype=float)
    if desc.n_ops > 0:
        K = np.array(_tree_matrix(desc, X, Y)[0])
        if Y is None and diag_add:
            K[np.diag_indices_from(K)] += float(diag_add)
        return K
    base = _leaf(desc, X.shape[1])(X, None if Y is None else np.asarray(Y, dtype=float))
    K = float(desc.amplitude) * base
    if Y is None:
        K[np.diag_indices_from(K)] += float(desc.white_noise)
    K += float(desc.additive_const)
    if Y is None and diag_add:
        K[np.diag_indices_from(K)] += float(diag_add)
    return K


class CpuMatrix:
    """Host stand-in for a ``gsum_mat``: the symmetric matrix, then its lower Cholesky factor."""

    def __init__(self, ctx, A):
        self._ctx = ctx
        self.A = np.array(A, dtype=float)
        self.n = self.A.shape[0]
        self.factored = False

    def to_host(self):
        return np.tril(self.A) if self.factored else self.A.copy()

    def scale_series(self, series, ref, ratio):
        """A_ij *= factor ref_i ref_j S(ratio_i ratio_j)  (TruncationProcess.cov, gsum/models.py:1343-1354)."""
        if self.factored:
            raise ValueError("scale_series needs an unfactored matrix")
        ref, ratio = np.asarray(ref, dtype=float), np.asarray(ratio, dtype=float)
        self.A *= _series_factor(series, ref, ratio, ref, ratio)

    def free(self):
        self.A = None


def _series_factor(series, ref_r, ratio_r, ref_c, ratio_c):
    end = np.inf if series.end < 0 else series.end
    exc = [series.excluded[i] for i in range(series.n_excluded)] or None
    S = geometric_sum(x=ratio_r[:, None] * ratio_c[None, :], start=series.start, end=end, excluded=exc)
    return series.factor * (ref_r[:, None] * ref_c[None, :]) * S


class CpuContext:
    """numpy / scipy implementation of the HipContext methods the model classes call."""

    device = None
    backend = "cpu"

    def __init__(self):
        self.calls = collections.Counter()
        self._X = self._Z = None
        self._options = {}

    # -- plumbing ------------------------------------------------------------------------------------
    def set_option(self, name, value):
        self._options[name] = int(value)

    def get_option(self, name):
        return self._options.get(name, 0)

    def close(self):
        pass

    @staticmethod
    def desc_array(descs):
        return list(descs)

    @staticmethod
    def tile_descs(unique, index):
        return [unique[int(t)] for t in index]

    # -- operator level ------------------------------------------------------------------------------
    def kernel_matrix(self, desc, X, Y=None, diag_add=0.0, series=None):
        self.calls["kernel_matrix"] += 1
        K = kernel_matrix(desc, X, Y, diag_add)
        if series is not None:
            sc, ref_x, ratio_x = series[0], np.asarray(series[1], float), np.asarray(series[2], float)
            ref_y, ratio_y = (ref_x, ratio_x) if Y is None else (np.asarray(series[3], float), np.asarray(series[4], float))
            K = _series_factor(sc, ref_x, ratio_x, ref_y, ratio_y) * K
        return K

    def kernel_matrix_dev(self, desc, X, diag_add=0.0):
        self.calls["kernel_matrix_dev"] += 1
        return CpuMatrix(self, kernel_matrix(desc, X, None, diag_add))

    def upload(self, A):
        A = np.asarray(A, dtype=float)
        if A.ndim != 2 or A.shape[0] != A.shape[1]:
            raise ValueError("square matrix expected")
        return CpuMatrix(self, A)

    def potrf(self, M: CpuMatrix) -> int:
        """numpy.linalg.cholesky's LAPACK call with its info (0 = success, k > 0: leading minor k is not positive definite)."""
        self.calls["potrf"] += 1
        if M.factored:
            raise ValueError("matrix is already factorised")
        c, info = dpotrf(M.A, lower=1, clean=1, overwrite_a=0)
        if info == 0:
            M.A = c
            M.factored = True
        return int(info)

    def factorize(self, desc, X, diag_add=0.0, series=None):
        K = self.kernel_matrix_dev(desc, X, diag_add=diag_add)
        if series is not None:
            K.scale_series(*series)
        return K, self.potrf(K)

    def forward_solve(self, L: CpuMatrix, rhs):
        rhs = np.asarray(rhs, dtype=float)
        return solve_triangular(L.A, rhs, lower=True)

    def forward_gram(self, L: CpuMatrix, rhs):
        self.calls["forward_gram"] += 1
        rhs = np.asarray(rhs, dtype=float)
        if rhs.ndim == 1:
            rhs = rhs[:, None]
        W = solve_triangular(L.A, rhs, lower=True)
        L.solved_W = W                              # (what the device factor's border rows hold: gsum_predict_var reads it)
        return W.T @ W, float(np.log(np.diag(L.A)).sum())

    def predict_var(self, L: CpuMatrix, desc, X, Xs, want_vtw=False):
        """gsum_predict_var's contract: column sums of squares of V = L^-1 kernel(X, Xs) and, with ``want_vtw``, V^T W as m x GSUM_MAX_RHS
        (columns beyond the k held right-hand sides zero) for the right-hand sides of the last forward_gram on this factor."""
        W = getattr(L, "solved_W", None)
        if want_vtw and W is None:
            raise ValueError("gsum_predict_var: V^T W needs right-hand sides solved against this factor first (gsum_forward_gram)")
        V = solve_triangular(L.A, kernel_matrix(desc, np.asarray(X, dtype=float), np.asarray(Xs, dtype=float)), lower=True)
        css = np.einsum("ij,ij->j", V, V)
        if not want_vtw:
            return css, None
        out = np.zeros((V.shape[1], GSUM_MAX_RHS))
        out[:, :W.shape[1]] = V.T @ W
        return css, out

    def cho_solve(self, L: CpuMatrix, B):
        return _cho_solve((L.A, True), np.asarray(B, dtype=float))

    def tri_multiply(self, L: CpuMatrix, Z):
        return np.tril(L.A) @ np.asarray(Z, dtype=float)

    def predict_terms(self, L: CpuMatrix, desc, X, Xs, rhs=None, want_cov=False, series=None):
        self.calls["predict_terms"] += 1
        X, Xs = np.asarray(X, dtype=float), np.asarray(Xs, dtype=float)
        Kon = kernel_matrix(desc, X, Xs)                             # n x m, two-argument form: no white noise
        if series is not None:
            sc, ref_x, ratio_x, ref_s, ratio_s = series
            Kon = Kon * _series_factor(sc, np.asarray(ref_x, float), np.asarray(ratio_x, float), np.asarray(ref_s, float),
                                       np.asarray(ratio_s, float))
        V = solve_triangular(L.A, Kon, lower=True)
        colsumsq = np.einsum("ij,ij->j", V, V)
        VtW = None
        if rhs is not None:
            rhs = np.asarray(rhs, dtype=float)
            if rhs.ndim == 1:
                rhs = rhs[:, None]
            VtW = V.T @ solve_triangular(L.A, rhs, lower=True)
        return colsumsq, VtW, (V.T @ V if want_cov else None)

    # -- fused hot path ------------------------------------------------------------------------------
    def _evaluate(self, desc, X, rhs, nugget):
        k = rhs.shape[1]
        M = CpuMatrix(self, kernel_matrix(desc, X, None, nugget))
        info = self.potrf(M)
        if info:
            return np.full((k, k), np.nan), np.nan, info, None
        G, sld = self.forward_gram(M, rhs)
        return G, sld, 0, M

    def lml_batch(self, descs, X, rhs, nugget):
        X, rhs = np.asarray(X, dtype=float), np.asarray(rhs, dtype=float)
        if rhs.shape[1] > GSUM_MAX_RHS:
            raise ValueError("k must be 0..GSUM_MAX_RHS")
        out = [self._evaluate(d, X, rhs, nugget)[:3] for d in descs]
        return self._stacked(out, rhs.shape[1])

    @staticmethod
    def _stacked(out, k):
        """(G, sld, info) of a batch as arrays -- of the right shapes for an empty batch too."""
        return (np.array([o[0] for o in out], dtype=float).reshape(len(out), k, k), np.array([o[1] for o in out], dtype=float),
                np.array([o[2] for o in out], dtype=np.int64))

    def set_inputs(self, X, rhs):
        self._X, self._Z = np.array(X, dtype=float), np.array(rhs, dtype=float)

    def set_inputs_sets(self, X, rhs_sets):
        self._X, self._Zsets = np.array(X, dtype=float), np.array(rhs_sets, dtype=float)
        self._Z = self._Zsets[0]

    def lml_resident_sets(self, descs, set_of, nugget):
        out = [self._evaluate(d, self._X, self._Zsets[int(s)], nugget)[:3] for d, s in zip(descs, set_of)]
        return self._stacked(out, self._Zsets.shape[2])

    def resident_shape(self):
        if self._X is None:
            return 0, 0, 0
        return self._X.shape[0], self._X.shape[1], self._Z.shape[1]

    def lml_resident(self, descs, nugget):
        if self._X is None:
            raise ValueError("gsum_set_inputs has not been called")
        return self.lml_batch(descs, self._X, self._Z, nugget)

    def lml_grad(self, desc, params, X, rhs, nugget):
        """(G, sld, info, trace (P,), H (P, k, k)): tr(R^-1 dR_p) and V^T dR_p V with V = R^-1 rhs (gsum/models.py:1041-1056)."""
        X, rhs = np.asarray(X, dtype=float), np.asarray(rhs, dtype=float)
        k, P = rhs.shape[1], len(params)
        G, sld, info, M = self._evaluate(desc, X, rhs, nugget)
        if info:
            return G, sld, info, np.zeros(P), np.zeros((P, k, k))
        n, d = X.shape
        V = _cho_solve((M.A, True), rhs)
        Rinv = _cho_solve((M.A, True), np.eye(n))
        dK = None
        if desc.n_ops == 0:
            term = ConstantKernel(float(desc.amplitude), constant_value_bounds=(1e-300, 1e300)) * _leaf(desc, d, free=True)
            _, dK = term(X, eval_gradient=True)              # [:, :, 0]: d / d log amplitude; [:, :, 1:]: length scale(s)
        trace, H = np.empty(P), np.empty((P, k, k))
        for p, pr in enumerate(params):
            if pr.code >= GradParam.TREE_CONST:
                dR = _tree_matrix(desc, X, None, pr)[1]
            elif pr.code == GradParam.AMPLITUDE:
                dR = dK[:, :, 0]
            elif pr.code == GradParam.LENGTH_ISO:
                dR = dK[:, :, 1]
            elif pr.code == GradParam.LENGTH_DIM:
                dR = dK[:, :, 1 + pr.dim]
            elif pr.code == GradParam.WHITE:
                dR = pr.weight * np.eye(n)
            else:
                dR = np.full((n, n), pr.weight)
            trace[p] = np.einsum("ij,ji->", Rinv, dR)
            H[p] = V.T @ dR @ V
        return G, sld, 0, trace, H

    def lml_grad_batch(self, descs, params, X, rhs, nugget):
        out = [self.lml_grad(d, p, X, rhs, nugget) for d, p in zip(descs, params)]
        return (np.array([o[0] for o in out]), np.array([o[1] for o in out]), np.array([o[2] for o in out], dtype=np.int64),
                np.array([o[3] for o in out]), np.array([o[4] for o in out]))


_cpu_ctx = None


def cpu_context() -> CpuContext:
    global _cpu_ctx
    if _cpu_ctx is None:
        _cpu_ctx = CpuContext()
    return _cpu_ctx


class CpuGroup:
    """``backend='cpu'`` counterpart of :class:`gsum_amd._lib.HipGroup`: ``world`` CPU contexts behind the same fan-out interface
    (``contexts``, ``map``, ``lml_resident``, ``lml_batch``, ``allgather``), so the sharding logic of ``devices=`` runs -- and is
    tested -- without a GPU.  Every member counts its own calls."""

    def __init__(self, world: int):
        if world < 1:
            raise ValueError("a group needs at least one member")
        self.contexts = [CpuContext() for _ in range(int(world))]
        self.devices = list(range(int(world)))
        self.gathers = 0

    def __len__(self):
        return len(self.contexts)

    def map(self, fn):
        import threading
        world = len(self.contexts)
        out, errs = [None] * world, [None] * world

        def run(r):
            try:
                out[r] = fn(r, self.contexts[r])
            except BaseException as exc:        # noqa: BLE001 -- re-raised on the calling thread
                errs[r] = exc
        threads = [threading.Thread(target=run, args=(r,)) for r in range(1, world)]
        for t in threads:
            t.start()
        run(0)
        for t in threads:
            t.join()
        for e in errs:
            if e is not None:
                raise e
        return out

    def set_inputs(self, X, rhs):
        for c in self.contexts:
            c.set_inputs(X, rhs)

    def _sharded(self, call, descs, k):
        from .grid import shard_range
        nk, world = len(descs), len(self.contexts)
        G, sld, info = np.full((nk, k, k), np.nan), np.full(nk, np.nan), np.full(nk, -1, dtype=np.int64)

        def block(r, ctx):
            lo, hi = shard_range(nk, r, world)
            if hi > lo:
                G[lo:hi], sld[lo:hi], info[lo:hi] = call(ctx, list(descs[lo:hi]))
        self.map(block)
        return G, sld, info

    def set_inputs_sets(self, X, rhs_sets):
        for c in self.contexts:
            c.set_inputs_sets(X, rhs_sets)

    def lml_resident_sets(self, descs, set_of, nugget, gather="host"):
        from .grid import shard_range
        k = self.contexts[0].resident_shape()[2]
        nk, world = len(descs), len(self.contexts)
        G, sld, info = np.full((nk, k, k), np.nan), np.full(nk, np.nan), np.full(nk, -1, dtype=np.int64)

        def block(r, ctx):
            lo, hi = shard_range(nk, r, world)
            if hi > lo:
                G[lo:hi], sld[lo:hi], info[lo:hi] = ctx.lml_resident_sets(list(descs[lo:hi]), list(set_of[lo:hi]), nugget)
        self.map(block)
        return self._gathered((G, sld, info), gather)

    def lml_resident(self, descs, nugget, gather="host"):
        k = self.contexts[0].resident_shape()[2]
        out = self._sharded(lambda ctx, part: ctx.lml_resident(part, nugget), descs, k)
        return self._gathered(out, gather)

    def lml_batch(self, descs, X, rhs, nugget, gather="host"):
        out = self._sharded(lambda ctx, part: ctx.lml_batch(part, X, rhs, nugget), descs, np.shape(rhs)[1])
        return self._gathered(out, gather)

    def _gathered(self, out, gather):
        if gather not in ("host", "rccl"):
            raise KeyError(gather)
        if gather == "rccl":
            self.gathers += 1                   # no devices to exchange between: the host arrays already are the gathered grid
        return out

    def allgather(self, a, rows=None):
        self.gathers += 1
        return np.array(a, dtype=np.float64)

    def get(self, name):
        return {"rccl": 0, "rccl_gathers": self.gathers, "devices_used": len(self.contexts)}.get(name, -1)


_cpu_groups = {}


def cpu_group(world: int) -> CpuGroup:
    g = _cpu_groups.get(int(world))
    if g is None:
        g = _cpu_groups[int(world)] = CpuGroup(world)
    return g
