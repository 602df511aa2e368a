"""ctypes binding of libgsum_hip.so (C ABI: include/gsum_hip.h).

There is no CPU fallback: if the shared library is missing or no MI355X is
visible, creating a :class:`HipContext` raises.  Build the library with
``python -m gsum_amd.build`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


LIB_PATH = os.environ.get("GSUM_HIP_LIBRARY") or os.path.join(_HERE, "libgsum_hip.so")
LAB_LIB_PATH = os.path.join(_HERE, "libgsum_hip_lab.so")     # the same sources with -DGSUM_LAB: + include/gsum_hip_debug.h

GSUM_MAX_D = 8
GSUM_MAX_RHS = 16
GSUM_MAX_LEAVES = 4
GSUM_MAX_OPS = 24
FAMILY = {"rbf": 0, "matern52": 1, "matern32": 2, "matern12": 3, "rq": 4, "expsine": 5, "matern_inf": 6, "dot": 7}
OP_ADD, OP_MUL, OP_LEAF, OP_CONST, OP_WHITE, OP_POW = 1, 2, 16, 32, 64, 128


class KernelLeaf(C.Structure):
    """Mirror of ``gsum_kernel_leaf``: one stationary leaf of a kernel tree."""
    _fields_ = [
        ("family", C.c_int32),
        ("anisotropic", C.c_int32),
        ("length_scale", C.c_double * GSUM_MAX_D),
        ("alpha", C.c_double),
    ]


class KernelDesc(C.Structure):
    """Mirror of ``gsum_kernel_desc``: the flattened form (``n_ops == 0``) or a postfix program over a Sum / Product tree."""
    _fields_ = [
        ("family", C.c_int32),
        ("anisotropic", C.c_int32),
        ("length_scale", C.c_double * GSUM_MAX_D),
        ("amplitude", C.c_double),
        ("additive_const", C.c_double),
        ("white_noise", C.c_double),
        ("n_ops", C.c_int32),
        ("n_leaves", C.c_int32),
        ("op", C.c_int32 * GSUM_MAX_OPS),
        ("cval", C.c_double * GSUM_MAX_OPS),
        ("leaf", KernelLeaf * GSUM_MAX_LEAVES),
    ]

    @property
    def is_tree(self):
        return self.n_ops > 0

    def one_arg_diagonal(self, X=None):
        """k(x, x) of the ONE-argument form (every stationary leaf exactly 1, WhiteKernel noise in): the diagonal of R_nn at
        gsum/models.py:824.  One number -- unless the tree holds a DotProduct leaf (x . x + sigma_0^2: kernels.py DotProduct.diag), then
        one value per row of ``X``."""
        if not self.is_tree:
            return self.amplitude * 1.0 + self.white_noise + self.additive_const
        stack = []
        for k in range(self.n_ops):
            op = self.op[k]
            if op >= OP_POW:
                stack.append(stack.pop() ** self.cval[op - OP_POW])          # Exponentiation.diag: kernel.diag(X) ** exponent
            elif op >= OP_CONST:
                stack.append(self.cval[op - (OP_WHITE if op >= OP_WHITE else OP_CONST)])
            elif op >= OP_LEAF:
                lf = self.leaf[op - OP_LEAF]
                if lf.family == FAMILY["dot"]:
                    if X is None:
                        raise ValueError("the diagonal of a kernel with a DotProduct leaf depends on the points: pass X")
                    Xa = np.asarray(X, dtype=float)
                    stack.append(np.einsum("ij,ij->i", Xa, Xa) + lf.length_scale[0] ** 2)
                else:
                    stack.append(1.0)
            else:
                b, a = stack.pop(), stack.pop()
                stack.append(a + b if op == OP_ADD else a * b)
        return stack[0]

    def without_white(self):
        """A copy whose WhiteKernel terms are zero: the kernel called with BOTH arguments given (kernels.py:1413-1414)."""
        import copy
        out = copy.copy(self)
        out.white_noise = 0.0
        for k in range(self.n_ops):
            if OP_WHITE <= self.op[k] < OP_POW:
                out.cval[self.op[k] - OP_WHITE] = 0.0
        return out

    def plus_constant(self, c):
        """A copy with an additive constant on top of the whole kernel (ConjugateStudentProcess: corr + basis disp basis^T)."""
        import copy
        out = copy.copy(self)
        if not self.is_tree:
            out.additive_const += float(c)
            return out
        used = {self.op[k] - (OP_POW if self.op[k] >= OP_POW else OP_WHITE if self.op[k] >= OP_WHITE else OP_CONST)
                for k in range(self.n_ops) if self.op[k] >= OP_CONST}
        slot = next(i for i in range(GSUM_MAX_OPS) if i not in used)
        if self.n_ops + 2 > GSUM_MAX_OPS:
            raise NotImplementedError("kernel tree too large for the device descriptor")
        out.cval[slot] = float(c)
        out.op[self.n_ops] = OP_CONST + slot
        out.op[self.n_ops + 1] = OP_ADD
        out.n_ops = self.n_ops + 2
        return out

    def __repr__(self):
        if self.is_tree:
            return f"KernelDesc(tree: ops={list(self.op)[:self.n_ops]}, leaves={self.n_leaves})"
        ls = list(self.length_scale)[: (GSUM_MAX_D if self.anisotropic else 1)]
        return (f"KernelDesc(family={self.family}, aniso={self.anisotropic}, ls={ls}, amp={self.amplitude}, "
                f"add={self.additive_const}, white={self.white_noise})")


class SeriesScale(C.Structure):
    """Mirror of ``gsum_series_scale``: cov_ij = factor ref_i ref_j S(ratio_i ratio_j) kernel_ij."""
    _fields_ = [
        ("start", C.c_int32),
        ("end", C.c_int32),
        ("n_excluded", C.c_int32),
        ("excluded", C.c_int32 * 16),
        ("factor", C.c_double),
    ]

    @classmethod
    def make(cls, start, end, excluded=None, factor=1.0):
        """``end`` may be ``numpy.inf`` (infinite sum); ``excluded`` is an int, a sequence or None (helpers.py:149-182)."""
        sc = cls()
        if end < start:
            raise ValueError('end must be greater than or equal to start')          # helpers.py:175-176
        sc.start = int(start)
        sc.end = -1 if np.isinf(end) else int(end)
        exc = [] if excluded is None else [int(e) for e in np.atleast_1d(excluded)]
        if len(exc) > 16:
            raise ValueError("at most 16 excluded orders")
        sc.n_excluded = len(exc)
        for i, e in enumerate(exc):
            sc.excluded[i] = e
        sc.factor = float(factor)
        return sc


class GradParam(C.Structure):
    """Mirror of ``gsum_grad_param``: which derivative d kernel(X) / d theta_p is."""
    _fields_ = [("code", C.c_int32), ("dim", C.c_int32), ("weight", C.c_double)]

    AMPLITUDE, LENGTH_ISO, LENGTH_DIM, WHITE, ADDITIVE = range(5)
    TREE_CONST, TREE_WHITE, TREE_LENGTH_ISO, TREE_LENGTH_DIM, TREE_ALPHA = range(16, 21)     # parameters of a kernel tree (see gsum_hip.h)

    def __repr__(self):
        return f"GradParam(code={self.code}, dim={self.dim}, weight={self.weight})"


_p = C.c_void_p
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_kp = C.POINTER(KernelDesc)

# name -> (restype, argtypes); must list every symbol include/gsum_hip.h declares (PROTOTYPES) / include/gsum_hip_debug.h adds
# in the lab build (LAB_PROTOTYPES)
PROTOTYPES = {
    "gsum_init": (C.c_int, [C.c_int, C.POINTER(_p)]),
    "gsum_destroy": (None, [_p]),
    "gsum_last_error": (C.c_char_p, [_p]),
    "gsum_set_option": (C.c_int, [_p, C.c_char_p, C.c_int64]),
    "gsum_get_option": (C.c_int64, [_p, C.c_char_p]),
    "gsum_kernel_build": (C.c_int, [_p, _kp, _dp, C.c_int64, C.c_int32, _dp, C.c_int64, C.c_double, _dp]),
    "gsum_kernel_build_series": (C.c_int, [_p, _kp, _dp, C.c_int64, C.c_int32, _dp, C.c_int64, C.c_double, C.POINTER(SeriesScale),
                                          _dp, _dp, _dp, _dp, _dp]),
    "gsum_kernel_build_dev": (C.c_int, [_p, _kp, _dp, C.c_int64, C.c_int32, C.c_double, C.POINTER(_p)]),
    "gsum_mat_from_host": (C.c_int, [_p, _dp, C.c_int64, C.POINTER(_p)]),
    "gsum_potrf_lower": (C.c_int, [_p, _p, _ip]),
    "gsum_forward_gram": (C.c_int, [_p, _p, _dp, C.c_int64, C.c_int32, _dp, _dp]),
    "gsum_forward_solve": (C.c_int, [_p, _p, _dp, C.c_int64, C.c_int32, _dp]),
    "gsum_cho_solve": (C.c_int, [_p, _p, _dp, C.c_int64, C.c_int32, _dp]),
    "gsum_predict_terms": (C.c_int, [_p, _p, _kp, _dp, C.c_int64, C.c_int32, _dp, C.c_int64, _dp, C.c_int32,
                                     _dp, _dp, _dp]),
    "gsum_predict_var": (C.c_int, [_p, _p, _kp, _dp, C.c_int64, C.c_int32, _dp, C.c_int64, _dp, _dp]),
    "gsum_tri_multiply": (C.c_int, [_p, _p, _dp, C.c_int64, C.c_int32, _dp]),
    "gsum_mat_scale_series": (C.c_int, [_p, _p, C.POINTER(SeriesScale), _dp, _dp]),
    "gsum_predict_terms_series": (C.c_int, [_p, _p, _kp, _dp, C.c_int64, C.c_int32, _dp, C.c_int64, _dp, C.c_int32,
                                            C.POINTER(SeriesScale), _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "gsum_mat_to_host": (C.c_int, [_p, _p, _dp]),
    "gsum_mat_n": (C.c_int64, [_p]),
    "gsum_mat_free": (None, [_p, _p]),
    "gsum_lml_batch": (C.c_int, [_p, _kp, C.c_int32, _dp, C.c_int64, C.c_int32, _dp, C.c_int32, C.c_double,
                                 _dp, _dp, _ip]),
    "gsum_lml_grad": (C.c_int, [_p, _kp, C.POINTER(GradParam), C.c_int32, _dp, C.c_int64, C.c_int32, _dp, C.c_int32,
                                C.c_double, _dp, _dp, _ip, _dp, _dp]),
    "gsum_lml_grad_batch": (C.c_int, [_p, _kp, C.c_int32, C.POINTER(GradParam), C.c_int32, _dp, C.c_int64, C.c_int32, _dp, C.c_int32,
                                C.c_double, _dp, _dp, _ip, _dp, _dp]),
    "gsum_set_inputs": (C.c_int, [_p, _dp, C.c_int64, C.c_int32, _dp, C.c_int32]),
    "gsum_resident_shape": (C.c_int, [_p, _ip, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gsum_lml_resident": (C.c_int, [_p, _kp, C.c_int32, C.c_double, _dp, _dp, _ip]),
    "gsum_set_inputs_sets": (C.c_int, [_p, _dp, C.c_int64, C.c_int32, _dp, C.c_int32, C.c_int32]),
    "gsum_lml_resident_sets": (C.c_int, [_p, _kp, C.POINTER(C.c_int32), C.c_int32, C.c_double, _dp, _dp, _ip]),
    "gsum_shard_range": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, _ip, _ip]),
    "gsum_lml_resident_shard": (C.c_int, [_p, _kp, C.c_int32, C.c_int32, C.c_int32, C.c_double, _dp, _dp, _ip, _ip, _ip]),
    "gsum_timers": (C.c_int, [_p, _dp, C.c_int32]),
    "gsum_kernel_profile": (C.c_int, [_p, _dp, _dp, _ip]),
    "gsum_init_multi": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(_p)]),
    "gsum_group_adopt": (C.c_int, [C.c_int, C.POINTER(_p), C.POINTER(_p)]),
    "gsum_group_destroy": (None, [_p]),
    "gsum_group_size": (C.c_int32, [_p]),
    "gsum_group_ctx": (_p, [_p, C.c_int32]),
    "gsum_group_last_error": (C.c_char_p, [_p]),
    "gsum_group_get": (C.c_int64, [_p, C.c_char_p]),
    "gsum_group_set_inputs": (C.c_int, [_p, _dp, C.c_int64, C.c_int32, _dp, C.c_int32]),
    "gsum_group_lml_resident": (C.c_int, [_p, _kp, C.c_int32, C.c_double, _dp, _dp, _ip, C.c_int32]),
    "gsum_group_set_inputs_sets": (C.c_int, [_p, _dp, C.c_int64, C.c_int32, _dp, C.c_int32, C.c_int32]),
    "gsum_group_lml_resident_sets": (C.c_int, [_p, _kp, C.POINTER(C.c_int32), C.c_int32, C.c_double, _dp, _dp, _ip, C.c_int32]),
    "gsum_lml_batch_multi": (C.c_int, [_p, _kp, C.c_int32, _dp, C.c_int64, C.c_int32, _dp, C.c_int32, C.c_double,
                                       _dp, _dp, _ip, C.c_int32]),
    "gsum_group_allgather": (C.c_int, [_p, _dp, C.c_int64, C.c_int64]),
}
LAB_PROTOTYPES = {
    "gsum_debug_pipe_probe": (C.c_int, [_p, C.c_int32, C.POINTER(C.c_int32)]),
    "gsum_debug_diag_stamps": (C.c_int, [_p, _ip]),
    "gsum_debug_chain_stamps": (C.c_int, [_p, _dp, C.c_int32, C.POINTER(C.c_int32)]),
    "gsum_probe_mfma_f64": (C.c_int, [_p, C.c_int32, C.c_int32, C.c_int32, _dp]),
    "gsum_probe_hbm_write": (C.c_int, [_p, C.c_int64, _dp]),
    "gsum_bench_gemm_nt": (C.c_int, [_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _dp]),
    "gsum_debug_gemm_nt": (C.c_int, [_p, C.c_int32, C.c_int32, _dp, _dp, _dp, C.c_int64, C.c_int64, C.c_int64,
                                     C.c_int32, C.c_double]),
}

_lib = None
_lab_lib = None
_lib_lock = threading.Lock()


def load_library(path: str | None = None, lab: bool = False):
    """dlopen libgsum_hip.so (``lab=True``: libgsum_hip_lab.so, the lab build) and attach prototypes.  Raises if it is absent."""
    global _lib, _lab_lib
    with _lib_lock:
        if path is None and (_lab_lib if lab else _lib) is not None:
            return _lab_lib if lab else _lib
        p = path or (LAB_LIB_PATH if lab else LIB_PATH)
        if not os.path.exists(p):
            raise RuntimeError(
                f"{p} not found: the HIP extension is not built. Run `python -m gsum_amd.build" + (" --lab" if lab else "") + "` "
                "(hipcc --offload-arch=gfx950). The 'hip' backend has no CPU fallback.")
        lib = C.CDLL(p)
        protos = dict(PROTOTYPES, **LAB_PROTOTYPES) if lab else PROTOTYPES
        for name, (res, args) in protos.items():
            fn = getattr(lib, name)        # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if path is None:
            if lab:
                _lab_lib = lib
            else:
                _lib = lib
        return lib


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def _ptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


_pipes_warned = False
_live_contexts = weakref.WeakSet()        # every HipContext / HipGroup alive: closed at exit, groups first (see _close_default_contexts)
_live_groups = weakref.WeakSet()


def _warn_pipes(ctx):
    """Once per process: gsum_init's pairwise stream probe found two of the context's streams taking turns."""
    global _pipes_warned
    if _pipes_warned or int(ctx._lib.gsum_get_option(ctx._h, b"pipes_ok")) != 0:
        return
    _pipes_warned = True
    import warnings
    pm = int(ctx._lib.gsum_get_option(ctx._h, b"pipe_overlap_permille"))
    warnings.warn(f"gsum_amd: two of the context's four HIP streams on device {ctx.device} do not run side by side (overlap {pm / 10:.0f} % "
                  "of a 100-us probe kernel) although gsum_init replaced the offending stream eight times: a tool serialises dispatches, "
                  "or the process holds more streams than the GPU has hardware queues.  Results are unaffected; batches run 3-6 % and single factorisations 30-70 % slower (DESIGN.md section 4.1).", RuntimeWarning, stacklevel=3)


class ChainAborted(RuntimeError):
    """gsum_potrf_lower returned GSUM_ERR_CHAIN_ABORT: the persistent-chain schedule of a single factorisation timed out (streams of
    this process do not run side by side -- a tool that serialises dispatches); the matrix is destroyed, the schedule is now off for
    the context.  Rebuild the matrix and factorise again (``HipContext.factorize`` does)."""


class DeviceMatrix:
    """Owner of a ``gsum_mat*`` (device-resident symmetric matrix or Cholesky factor)."""

    def __init__(self, ctx: "HipContext", handle):
        self._ctx = ctx
        self._h = handle
        self.n = int(ctx._lib.gsum_mat_n(handle))
        self.factored = False

    def to_host(self) -> np.ndarray:
        out = np.empty((self.n, self.n))
        self._ctx._check(self._ctx._lib.gsum_mat_to_host(self._ctx._h, self._h, _ptr(out)))
        return out

    def scale_series(self, series: "SeriesScale", ref, ratio):
        """In place: A_ij *= factor ref_i ref_j S(ratio_i ratio_j) (TruncationProcess.cov, models.py:1343-1354)."""
        ref, ratio = _f64(ref), _f64(ratio)
        if ref.shape != (self.n,) or ratio.shape != (self.n,):
            raise ValueError("ref and ratio must have one value per row")
        self._ctx._check(self._ctx._lib.gsum_mat_scale_series(self._ctx._h, self._h, C.byref(series), _ptr(ref), _ptr(ratio)))

    def free(self):
        if self._h is not None and self._ctx._h is not None:
            self._ctx._lib.gsum_mat_free(self._ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HipContext:
    """One GPU, one ``gsum_ctx``.  Not thread-safe (one context per thread)."""

    def __init__(self, device: int = 0, lab: bool = False):
        self._lib = load_library(lab=lab)
        self.lab = bool(lab)
        h = _p()
        rc = self._lib.gsum_init(int(device), C.byref(h))
        if rc != 0:
            msg = self._lib.gsum_last_error(None)
            raise RuntimeError(f"gsum_init(device={device}) failed: {msg.decode() if msg else rc}")
        self._h = h
        self.device = int(device)
        _live_contexts.add(self)
        _warn_pipes(self)

    # -- plumbing ------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            msg = self._lib.gsum_last_error(self._h)
            text = msg.decode() if msg else f"error {rc}"
            if rc == -2:
                raise ValueError(text)
            if rc == -3:
                raise ChainAborted(text)
            raise RuntimeError(text)

    def close(self):
        if getattr(self, "_h", None) is not None:
            if not getattr(self, "_borrowed", False):        # a member of an owning HipGroup is destroyed with its group
                self._lib.gsum_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int):
        self._check(self._lib.gsum_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = int(self._lib.gsum_get_option(self._h, name.encode()))
        if v < 0 and name not in ("chain_persist", "chain_probe", "medium_min_batch", "wave_shift", "pipes_ok", "pipe_overlap_permille", "pipe_heals"):
            raise ValueError(f"unknown option: {name}")
        return v

    # -- operator level ------------------------------------------------------
    def kernel_matrix(self, desc: KernelDesc, X, Y=None, diag_add: float = 0.0, series=None) -> np.ndarray:
        """kernel(X[, Y]) as a host array.  ``series`` = (SeriesScale, ref_x, ratio_x[, ref_y, ratio_y]) scales it on the device
        like TruncationProcess.cov before it is copied out (gsum_kernel_build_series)."""
        X = _f64(X)
        n, d = X.shape
        if series is not None:
            sc, ref_x, ratio_x = series[0], _f64(series[1]), _f64(series[2])
            if ref_x.shape != (n,) or ratio_x.shape != (n,):
                raise ValueError("series scaling needs one ref / ratio value per point")
            if Y is None:
                out, Yp, m, ref_y, ratio_y = np.empty((n, n)), None, 0, None, None
            else:
                Y = _f64(Y)
                if Y.shape[1] != d:
                    raise ValueError("X and Y must have the same number of features")
                m = Y.shape[0]
                ref_y, ratio_y = _f64(series[3]), _f64(series[4])
                if ref_y.shape != (m,) or ratio_y.shape != (m,):
                    raise ValueError("series scaling needs one ref / ratio value per point")
                out, Yp = np.empty((n, m)), _ptr(Y)
            self._check(self._lib.gsum_kernel_build_series(self._h, C.byref(desc), _ptr(X), n, d, Yp, m, float(diag_add), C.byref(sc),
                                                           _ptr(ref_x), _ptr(ratio_x), _ptr(ref_y), _ptr(ratio_y), _ptr(out)))
            return out
        if Y is None:
            out = np.empty((n, n))
            self._check(self._lib.gsum_kernel_build(self._h, C.byref(desc), _ptr(X), n, d, None, 0,
                                                    float(diag_add), _ptr(out)))
        else:
            Y = _f64(Y)
            if Y.shape[1] != d:
                raise ValueError("X and Y must have the same number of features")
            out = np.empty((n, Y.shape[0]))
            self._check(self._lib.gsum_kernel_build(self._h, C.byref(desc), _ptr(X), n, d, _ptr(Y), Y.shape[0],
                                                    0.0, _ptr(out)))
        return out

    def kernel_matrix_dev(self, desc: KernelDesc, X, diag_add: float = 0.0) -> DeviceMatrix:
        X = _f64(X)
        h = _p()
        self._check(self._lib.gsum_kernel_build_dev(self._h, C.byref(desc), _ptr(X), X.shape[0], X.shape[1],
                                                    float(diag_add), C.byref(h)))
        return DeviceMatrix(self, h)

    def upload(self, A) -> DeviceMatrix:
        A = _f64(A)
        if A.ndim != 2 or A.shape[0] != A.shape[1]:
            raise ValueError("square matrix expected")
        h = _p()
        self._check(self._lib.gsum_mat_from_host(self._h, _ptr(A), A.shape[0], C.byref(h)))
        return DeviceMatrix(self, h)

    def potrf(self, A: DeviceMatrix) -> int:
        """In-place lower Cholesky; returns LAPACK ``info`` (0 = success)."""
        info = C.c_int64(0)
        self._check(self._lib.gsum_potrf_lower(self._h, A._h, C.byref(info)))
        A.factored = info.value == 0
        return int(info.value)

    def factorize(self, desc: KernelDesc, X, diag_add: float = 0.0, series=None):
        """kernel(X) + diag_add I on the device [scaled like TruncationProcess.cov: ``series`` = (SeriesScale, ref, ratio)], then
        its Cholesky factor in place: ``(matrix, info)``.  What fit / predict / the factor-reuse grid do at models.py:708-711, 807-809.
        Should the single-factorisation schedule give up (``ChainAborted``: the matrix is destroyed, the library has switched to the
        host-enqueued schedule) the matrix is rebuilt and factorised once more, transparently -- the fused evaluation path already
        re-runs itself the same way."""
        for attempt in (0, 1):
            K = self.kernel_matrix_dev(desc, X, diag_add=diag_add)
            try:
                if series is not None:
                    K.scale_series(*series)
                return K, self.potrf(K)
            except ChainAborted:
                K.free()
                if attempt:
                    raise
            except BaseException:
                K.free()
                raise

    def forward_gram(self, L: DeviceMatrix, rhs):
        rhs = _f64(rhs)
        if rhs.ndim == 1:
            rhs = rhs[:, None]
        n, k = rhs.shape
        G = np.empty((k, k))
        sld = C.c_double(0.0)
        self._check(self._lib.gsum_forward_gram(self._h, L._h, _ptr(rhs), n, k, _ptr(G), C.byref(sld)))
        return G, float(sld.value)

    def forward_solve(self, L: DeviceMatrix, rhs) -> np.ndarray:
        """W = L^-1 rhs (n x k), k <= 16."""
        rhs = _f64(rhs)
        squeeze = rhs.ndim == 1
        if squeeze:
            rhs = rhs[:, None]
        n, k = rhs.shape
        W = np.empty((n, k))
        self._check(self._lib.gsum_forward_solve(self._h, L._h, _ptr(rhs), n, k, _ptr(W)))
        return W[:, 0] if squeeze else W

    def cho_solve(self, L: DeviceMatrix, B) -> np.ndarray:
        """scipy.linalg.cho_solve((L, True), B) on the device factor (any number of columns; 16 per device pass)."""
        B = _f64(B)
        squeeze = B.ndim == 1
        if squeeze:
            B = B[:, None]
        n, k = B.shape
        out = np.empty((n, k))
        for lo in range(0, k, GSUM_MAX_RHS):
            bc = np.ascontiguousarray(B[:, lo:lo + GSUM_MAX_RHS])
            xc = np.empty_like(bc)
            self._check(self._lib.gsum_cho_solve(self._h, L._h, _ptr(bc), n, bc.shape[1], _ptr(xc)))
            out[:, lo:lo + GSUM_MAX_RHS] = xc
        return out[:, 0] if squeeze else out

    def tri_multiply(self, L: DeviceMatrix, Z) -> np.ndarray:
        """L @ Z for a factorised matrix (any number of columns; 16 per device pass)."""
        Z = _f64(Z)
        squeeze = Z.ndim == 1
        if squeeze:
            Z = Z[:, None]
        n, k = Z.shape
        out = np.empty((n, k))
        for lo in range(0, k, GSUM_MAX_RHS):
            zc = np.ascontiguousarray(Z[:, lo:lo + GSUM_MAX_RHS])
            oc = np.empty_like(zc)
            self._check(self._lib.gsum_tri_multiply(self._h, L._h, _ptr(zc), n, zc.shape[1], _ptr(oc)))
            out[:, lo:lo + GSUM_MAX_RHS] = oc
        return out[:, 0] if squeeze else out

    def predict_terms(self, L: DeviceMatrix, desc: KernelDesc, X, Xs, rhs=None, want_cov=False, series=None):
        """``series`` = (SeriesScale, ref_x, ratio_x, ref_s, ratio_s) scales the cross matrix like cov(X, Xs, start, end)."""
        X, Xs = _f64(X), _f64(Xs)
        n, d = X.shape
        m = Xs.shape[0]
        colsumsq = np.empty(m)
        k = 0
        VtW = None
        if rhs is not None:
            rhs = _f64(rhs)
            if rhs.ndim == 1:
                rhs = rhs[:, None]
            k = rhs.shape[1]
            VtW = np.empty((m, k))
        cov = np.empty((m, m)) if want_cov else None
        if series is None:
            self._check(self._lib.gsum_predict_terms(self._h, L._h, C.byref(desc), _ptr(X), n, d, _ptr(Xs), m,
                                                     _ptr(rhs), k, _ptr(colsumsq), _ptr(VtW), _ptr(cov)))
        else:
            sc, ref_x, ratio_x, ref_s, ratio_s = series
            ref_x, ratio_x, ref_s, ratio_s = _f64(ref_x), _f64(ratio_x), _f64(ref_s), _f64(ratio_s)
            if ref_x.shape != (n,) or ratio_x.shape != (n,) or ref_s.shape != (m,) or ratio_s.shape != (m,):
                raise ValueError("series scaling needs one ref / ratio value per point")
            self._check(self._lib.gsum_predict_terms_series(
                self._h, L._h, C.byref(desc), _ptr(X), n, d, _ptr(Xs), m, _ptr(rhs), k, C.byref(sc), _ptr(ref_x),
                _ptr(ratio_x), _ptr(ref_s), _ptr(ratio_s), _ptr(colsumsq), _ptr(VtW), _ptr(cov)))
        return colsumsq, VtW, cov

    def predict_var(self, L: DeviceMatrix, desc: KernelDesc, X, Xs, want_vtw: bool = False):
        """gsum_predict_var (SURVEY.md 8b's name): the column sums of squares of V = L^-1 kernel(X, Xs) and, with ``want_vtw``, V^T W as
        m x GSUM_MAX_RHS (columns beyond the k right-hand sides zero) for the right-hand sides whose forward solve the factor already
        holds (the last forward_gram / forward_solve / predict_terms on it): models.py:822-836 after fit, nothing handed over again."""
        X, Xs = _f64(X), _f64(Xs)
        n, d = X.shape
        m = Xs.shape[0]
        css = np.empty(m)
        vtw = np.empty((m, GSUM_MAX_RHS)) if want_vtw else None
        self._check(self._lib.gsum_predict_var(self._h, L._h, C.byref(desc), _ptr(X), n, d, _ptr(Xs), m, _ptr(css), _ptr(vtw)))
        return css, vtw

    # -- fused hot path ------------------------------------------------------
    @staticmethod
    def tile_descs(unique, index):
        """``[unique[t] for t in index]`` as ONE contiguous ``gsum_kernel_desc[]`` without a Python object per entry: a likelihood surface
        names each of its ~100 distinct kernels thousands of times (notebook :1457-1459), and joining 8000 x 712 B in Python cost as much
        as the device call.  The array shares the memory of a numpy gather (kept alive on the result)."""
        block = np.frombuffer(b"".join(map(bytes, unique)), dtype=np.uint8).reshape(len(unique), C.sizeof(KernelDesc))
        tiled = np.ascontiguousarray(block[np.asarray(index, dtype=np.intp)])
        arr = (KernelDesc * tiled.shape[0]).from_buffer(tiled)
        arr._keep = tiled
        return arr

    @staticmethod
    def _desc_array(descs):
        """Contiguous ``gsum_kernel_desc[]`` for the C ABI.  An array built earlier (``HipContext.desc_array``) passes through:
        for the fused small-n path the marshalling of thousands of descriptors was more than half of the call."""
        if isinstance(descs, C.Array) and getattr(descs, "_type_", None) is KernelDesc:
            return descs
        return (KernelDesc * len(descs)).from_buffer_copy(b"".join(map(bytes, descs)))

    desc_array = _desc_array

    def lml_batch(self, descs, X, rhs, nugget: float):
        """K build + Cholesky + Gram/log-det for each descriptor (host inputs)."""
        X, rhs = _f64(X), _f64(rhs)
        n, d = X.shape
        k = rhs.shape[1]
        nk = len(descs)
        G = np.empty((nk, k, k))
        sld = np.empty(nk)
        info = np.zeros(nk, dtype=np.int64)
        arr = self._desc_array(descs)
        self._check(self._lib.gsum_lml_batch(self._h, arr, nk, _ptr(X), n, d, _ptr(rhs), k, float(nugget),
                                             _ptr(G), _ptr(sld), info.ctypes.data_as(_ip)))
        return G, sld, info

    def lml_grad(self, desc: KernelDesc, params, X, rhs, nugget: float):
        """One evaluation with gradient pieces: (G (k,k), sld, info, trace (P,), H (P,k,k)); see gsum_lml_grad."""
        X, rhs = _f64(X), _f64(rhs)
        n, d = X.shape
        k = rhs.shape[1]
        P = len(params)
        arr = (GradParam * P)()
        for i, pr in enumerate(params):
            arr[i].code, arr[i].dim, arr[i].weight = pr.code, pr.dim, pr.weight
        G = np.empty((k, k))
        sld = np.empty(1)
        info = np.empty(1, dtype=np.int64)
        trace = np.empty(P)
        H = np.empty((P, k, k))
        self._check(self._lib.gsum_lml_grad(self._h, C.byref(desc), arr, P, _ptr(X), n, d, _ptr(rhs), k, float(nugget),
                                            _ptr(G), _ptr(sld), info.ctypes.data_as(_ip), _ptr(trace), _ptr(H)))
        return G, float(sld[0]), int(info[0]), trace, H

    def lml_grad_batch(self, descs, params, X, rhs, nugget: float):
        """Value + gradient pieces of several kernels with one hyperparameter structure on one set of inputs, pipelined on
        the device (gsum_lml_grad_batch): G (m, k, k), sld (m,), info (m,), trace (m, P), H (m, P, k, k).  ``params`` is a
        list of m parameter lists (``describe_gradient`` of every kernel: the weights depend on the hyperparameter values)."""
        X, rhs = _f64(X), _f64(rhs)
        n, d = X.shape
        k = rhs.shape[1]
        m = len(descs)
        if len(params) != m:
            raise ValueError("one parameter list per kernel")
        P = len(params[0])
        arr = (GradParam * (P * m))()
        for j, plist in enumerate(params):
            if len(plist) != P:
                raise ValueError("every kernel must have the same hyperparameter structure")
            for i, pr in enumerate(plist):
                arr[j * P + i].code, arr[j * P + i].dim, arr[j * P + i].weight = pr.code, pr.dim, pr.weight
        G = np.empty((m, k, k))
        sld = np.empty(m)
        info = np.empty(m, dtype=np.int64)
        trace = np.empty((m, P))
        H = np.empty((m, P, k, k))
        self._check(self._lib.gsum_lml_grad_batch(self._h, self._desc_array(descs), m, arr, P, _ptr(X), n, d, _ptr(rhs), k,
                                                  float(nugget), _ptr(G), _ptr(sld), info.ctypes.data_as(_ip), _ptr(trace), _ptr(H)))
        return G, sld, info, trace, H

    def set_inputs(self, X, rhs):
        X, rhs = _f64(X), _f64(rhs)
        self._check(self._lib.gsum_set_inputs(self._h, _ptr(X), X.shape[0], X.shape[1], _ptr(rhs), rhs.shape[1]))

    def set_inputs_sets(self, X, rhs_sets):
        """Several right-hand-side sets resident at once (``rhs_sets``: (n_sets, n, k)); see :meth:`lml_resident_sets`."""
        X, Z = _f64(X), _f64(rhs_sets)
        if Z.ndim != 3 or Z.shape[1] != X.shape[0]:
            raise ValueError("rhs_sets must have shape (n_sets, n, k)")
        self._check(self._lib.gsum_set_inputs_sets(self._h, _ptr(X), X.shape[0], X.shape[1], _ptr(Z), Z.shape[0], Z.shape[2]))

    def lml_resident_sets(self, descs, set_of, nugget: float):
        """``lml_resident`` with evaluation i reading right-hand-side set ``set_of[i]`` (gsum_lml_resident_sets): a whole
        (ratio, theta) surface in one call."""
        n, _, k = self.resident_shape()
        if n == 0:
            raise ValueError("gsum_set_inputs has not been called")
        nk = len(descs)
        sets = np.ascontiguousarray(set_of, dtype=np.int32)
        if sets.shape != (nk,):
            raise ValueError("one set index per descriptor")
        G, sld, info = np.empty((nk, k, k)), np.empty(nk), np.zeros(nk, dtype=np.int64)
        self._check(self._lib.gsum_lml_resident_sets(self._h, self._desc_array(descs), sets.ctypes.data_as(C.POINTER(C.c_int32)), nk,
                                                     float(nugget), _ptr(G), _ptr(sld), info.ctypes.data_as(_ip)))
        return G, sld, info

    def resident_shape(self):
        """(n, d, k) of the inputs gsum_set_inputs left on the device ((0, 0, 0) before the first call)."""
        n, d, k = C.c_int64(0), C.c_int32(0), C.c_int32(0)
        self._check(self._lib.gsum_resident_shape(self._h, C.byref(n), C.byref(d), C.byref(k)))
        return int(n.value), int(d.value), int(k.value)

    def lml_resident(self, descs, nugget: float):
        n, _, k = self.resident_shape()       # the library's own record, not a cached copy
        if n == 0:
            raise ValueError("gsum_set_inputs has not been called")
        nk = len(descs)
        G = np.empty((nk, k, k))
        sld = np.empty(nk)
        info = np.zeros(nk, dtype=np.int64)
        arr = self._desc_array(descs)
        self._check(self._lib.gsum_lml_resident(self._h, arr, nk, float(nugget), _ptr(G), _ptr(sld),
                                                info.ctypes.data_as(_ip)))
        return G, sld, info

    def lml_resident_shard(self, descs, nugget: float, rank: int, world: int):
        """This rank's slice of ``descs`` (gsum_shard_range), written to its positions of full-length arrays (NaN / -1
        elsewhere).  Returns (G, sld, info, lo, hi); the arrays are views of length ``len(descs)`` into buffers of
        ``world * ceil(len(descs) / world)`` leading entries, which is what an equal-chunk in-place all-gather
        (``ncclAllGather(buf + lo, buf, chunk, ...)``, include/gsum_hip.h) reads and writes: ``G.base`` etc. are those
        buffers."""
        n, _, k = self.resident_shape()
        if n == 0:
            raise ValueError("gsum_set_inputs has not been called")
        nk = len(descs)
        padded = int(world) * (-(-nk // int(world))) if world > 0 else nk
        G = np.full((padded, k, k), np.nan)[:nk]
        sld = np.full(padded, np.nan)[:nk]
        info = np.full(padded, -1, dtype=np.int64)[:nk]
        lo, hi = C.c_int64(0), C.c_int64(0)
        arr = self._desc_array(descs)
        self._check(self._lib.gsum_lml_resident_shard(self._h, arr, nk, int(rank), int(world), float(nugget), _ptr(G), _ptr(sld),
                                                      info.ctypes.data_as(_ip), C.byref(lo), C.byref(hi)))
        return G, sld, info, int(lo.value), int(hi.value)

    # -- measurement ---------------------------------------------------------
    def timers(self):
        ms = np.zeros(4)
        self._check(self._lib.gsum_timers(self._h, _ptr(ms), 4))
        return dict(build_ms=ms[0], potrf_ms=ms[1], finalize_ms=ms[2], total_ms=ms[3])

    def host_enqueue_ms(self) -> float:
        """Host wall time spent enqueuing the most recent evaluation (launch-overhead diagnostics)."""
        v = np.zeros(10)
        self._check(self._lib.gsum_timers(self._h, _ptr(v), 10))
        return float(v[9])

    def diag_stamps(self):
        """Shader-cycle stamps of the last diagonal-block kernel (needs option diag_stamps=1)."""
        v = np.zeros(9)
        self._check(self._lib.gsum_timers(self._h, _ptr(v), 9))
        tot, ticks = v[7], v[8]
        return dict(prologue_cyc=v[4], loop_cyc=v[5], inverse_cyc=v[6], total_cyc=tot,
                    total_us=ticks * 0.01, clock_ghz=(tot / ticks * 0.1) if ticks else 0.0)

    def diag_stamps_raw(self):
        v = np.zeros(64, dtype=np.int64)
        self._check(self._lib.gsum_debug_diag_stamps(self._h, v.ctypes.data_as(_ip)))
        return v

    def chain_stamps(self, max_steps: int = 512):
        """(steps, 24) array of the last persistent-chain factorisation's realtime stamps in microseconds from its
        start (NaN: not written); needs ``set_option("chain_stamps", 1)``.  Column meaning: include/gsum_hip.h."""
        out = np.zeros((max_steps, 24))
        steps = C.c_int32(0)
        self._check(self._lib.gsum_debug_chain_stamps(self._h, _ptr(out), max_steps, C.byref(steps)))
        out = out[: steps.value]
        return np.where(out < 0, np.nan, out * 0.01)

    PROFILE_CLASSES = ("kernel_build", "diag_block", "panel_gemm", "bulk_update", "other")

    def kernel_profile(self):
        """Per kernel class {name: dict(ms, flops, launches)} of the launches profiled since the last read-out (option
        "profile_gemm" = N: every launch of every N-th fused evaluation); resets the record."""
        ms, fl, cnt = np.zeros(5), np.zeros(5), np.zeros(5, dtype=np.int64)
        self._check(self._lib.gsum_kernel_profile(self._h, _ptr(ms), _ptr(fl), cnt.ctypes.data_as(_ip)))
        return {name: dict(ms=float(ms[i]), flops=float(fl[i]), launches=int(cnt[i]))
                for i, name in enumerate(self.PROFILE_CLASSES)}

    def pipe_probe(self, extra: int = 0) -> np.ndarray:
        """(4 + extra)^2 matrix of pairwise overlaps (fraction of a 100-us kernel) of the context's four streams and ``extra``
        streams created for the call (lab build)."""
        n = 4 + int(extra)
        out = (C.c_int32 * (n * n))()
        self._check(self._lib.gsum_debug_pipe_probe(self._h, int(extra), out))
        return np.array(out[:], dtype=float).reshape(n, n) / 1000.0

    def probe_mfma_f64(self, iters=10000, waves_per_simd=1, n_acc=8):
        """dict(tflops, cycles_per_mfma, clock_ghz) of a register-resident fp64 MFMA loop."""
        v = np.zeros(3)
        self._check(self._lib.gsum_probe_mfma_f64(self._h, iters, waves_per_simd, n_acc, _ptr(v)))
        return dict(tflops=float(v[0]), cycles_per_mfma=float(v[1]), clock_ghz=float(v[2]))

    def probe_hbm_write(self, nbytes=1 << 30) -> float:
        v = C.c_double(0)
        self._check(self._lib.gsum_probe_hbm_write(self._h, nbytes, C.byref(v)))
        return float(v.value)

    def bench_gemm_nt(self, cfg, M, N, K, tri=False, lda=None, reps=5):
        """(TFLOP/s, us per launch) of the MFMA tile kernel on device-resident random operands."""
        v = np.zeros(2)
        self._check(self._lib.gsum_bench_gemm_nt(self._h, cfg, int(tri), M, N, K, lda or K, reps, _ptr(v)))
        return float(v[0]), float(v[1])

    def debug_gemm_nt(self, cfg, Cm, A, B, tri=False, beta=1, sign=-1.0):
        Cm, A, B = _f64(Cm).copy(), _f64(A), _f64(B)
        M, K = A.shape
        N = B.shape[0]
        self._check(self._lib.gsum_debug_gemm_nt(self._h, cfg, int(tri), _ptr(Cm), _ptr(A), _ptr(B), M, N, K,
                                                 int(beta), float(sign)))
        return Cm


GATHER_FLAGS = {"host": 0, "rccl": 1}          # GSUM_GATHER_RCCL


class HipGroup:
    """Several GPUs from ONE process: a ``gsum_group`` over one context per device (include/gsum_hip.h, "multi-GPU from ONE caller
    thread").  ``devices``: a list of device indices, or "all".  By default the group adopts the process-wide contexts
    (``default_context(d)``), so a model fitted on device 0 and a group that includes device 0 share that context's streams and
    workspaces; ``own=True`` opens contexts of its own with ``gsum_init_multi``.  The sharded calls return arrays equal, bit for
    bit, to the one-device call; ``gather="rccl"`` additionally exchanges the result rows between the devices with an in-place
    ``ncclAllGather`` and checks every rank's copy against rank 0's."""

    def __init__(self, devices="all", lab: bool = False, own: bool = False):
        self._lib = load_library(lab=lab)
        self.lab = bool(lab)
        self._h = None
        h = _p()
        if devices is None:
            devices = "all"
        if own:
            ids = None if isinstance(devices, str) else [int(d) for d in devices]
            if isinstance(devices, str) and devices != "all":
                raise ValueError('devices must be "all" or a sequence of device indices')
            arr = (C.c_int * len(ids))(*ids) if ids else None
            rc = self._lib.gsum_init_multi(len(ids) if ids else 0, arr, C.byref(h))
            if rc != 0:
                msg = self._lib.gsum_group_last_error(None)
                raise RuntimeError(f"gsum_init_multi({devices}) failed: {msg.decode() if msg else rc}")
            self._h = h
            self.contexts = []
            for i in range(int(self._lib.gsum_group_size(h))):
                c = HipContext.__new__(HipContext)
                c._lib, c.lab, c._borrowed = self._lib, self.lab, True
                c._h = _p(self._lib.gsum_group_ctx(h, i))
                c.device = ids[i] if ids else i
                self.contexts.append(c)
                _warn_pipes(c)
        else:
            ids = resolve_devices(devices)
            if len(set(ids)) != len(ids):
                raise ValueError("a device may appear once in a group (own=True opens separate contexts on one device)")
            self.contexts = [(lab_context if lab else default_context)(d) for d in ids]
            arr = (_p * len(ids))(*[c._h for c in self.contexts])
            rc = self._lib.gsum_group_adopt(len(ids), arr, C.byref(h))
            if rc != 0:
                msg = self._lib.gsum_group_last_error(None)
                raise RuntimeError(f"gsum_group_adopt failed: {msg.decode() if msg else rc}")
            self._h = h
        self.devices = [c.device for c in self.contexts]
        _live_groups.add(self)

    def __len__(self):
        return len(self.contexts)

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.gsum_group_last_error(self._h)
            text = msg.decode() if msg else f"error {rc}"
            raise (ValueError if rc == -2 else RuntimeError)(text)

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._lib.gsum_group_destroy(self._h)
            self._h = None
            for c in self.contexts:
                if getattr(c, "_borrowed", False):
                    c._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get(self, name: str) -> int:
        return int(self._lib.gsum_group_get(self._h, name.encode()))

    def set_option(self, name: str, value: int):
        for c in self.contexts:
            c.set_option(name, value)

    def set_inputs(self, X, rhs):
        X, rhs = _f64(X), _f64(rhs)
        self._check(self._lib.gsum_group_set_inputs(self._h, _ptr(X), X.shape[0], X.shape[1], _ptr(rhs), rhs.shape[1]))

    def lml_resident(self, descs, nugget: float, gather: str = "host"):
        """``HipContext.lml_resident`` with the descriptors block-partitioned over the group's devices (gsum_shard_range)."""
        n, _, k = self.contexts[0].resident_shape()
        if n == 0:
            raise ValueError("gsum_group_set_inputs has not been called")
        nk = len(descs)
        G, sld, info = np.empty((nk, k, k)), np.empty(nk), np.zeros(nk, dtype=np.int64)
        self._check(self._lib.gsum_group_lml_resident(self._h, HipContext._desc_array(descs), nk, float(nugget), _ptr(G), _ptr(sld),
                                                      info.ctypes.data_as(_ip), GATHER_FLAGS[gather]))
        return G, sld, info

    def set_inputs_sets(self, X, rhs_sets):
        X, Z = _f64(X), _f64(rhs_sets)
        if Z.ndim != 3 or Z.shape[1] != X.shape[0]:
            raise ValueError("rhs_sets must have shape (n_sets, n, k)")
        self._check(self._lib.gsum_group_set_inputs_sets(self._h, _ptr(X), X.shape[0], X.shape[1], _ptr(Z), Z.shape[0], Z.shape[2]))

    def lml_resident_sets(self, descs, set_of, nugget: float, gather: str = "host"):
        """``HipContext.lml_resident_sets`` with the descriptors block-partitioned over the devices (every device holds all sets)."""
        n, _, k = self.contexts[0].resident_shape()
        if n == 0:
            raise ValueError("gsum_group_set_inputs_sets has not been called")
        nk = len(descs)
        sets = np.ascontiguousarray(set_of, dtype=np.int32)
        if sets.shape != (nk,):
            raise ValueError("one set index per descriptor")
        G, sld, info = np.empty((nk, k, k)), np.empty(nk), np.zeros(nk, dtype=np.int64)
        self._check(self._lib.gsum_group_lml_resident_sets(self._h, HipContext._desc_array(descs), sets.ctypes.data_as(C.POINTER(C.c_int32)), nk,
                                                           float(nugget), _ptr(G), _ptr(sld), info.ctypes.data_as(_ip), GATHER_FLAGS[gather]))
        return G, sld, info

    def lml_batch(self, descs, X, rhs, nugget: float, gather: str = "host"):
        """``HipContext.lml_batch`` over the group's devices (gsum_lml_batch_multi)."""
        X, rhs = _f64(X), _f64(rhs)
        n, d = X.shape
        k = rhs.shape[1]
        nk = len(descs)
        G, sld, info = np.empty((nk, k, k)), np.empty(nk), np.zeros(nk, dtype=np.int64)
        self._check(self._lib.gsum_lml_batch_multi(self._h, HipContext._desc_array(descs), nk, _ptr(X), n, d, _ptr(rhs), k, float(nugget),
                                                   _ptr(G), _ptr(sld), info.ctypes.data_as(_ip), GATHER_FLAGS[gather]))
        return G, sld, info

    def allgather(self, a, rows=None):
        """In-place RCCL all-gather of a row-partitioned float64 array (leading axis; rank r's rows are
        ``gsum_shard_range(rows, r, len(group))``) through the devices; returns rank 0's gathered copy, which the library has
        compared with every other rank's."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        rows = a.shape[0] if rows is None else int(rows)
        width = a.size // rows if rows else 1
        out = a.copy()
        self._check(self._lib.gsum_group_allgather(self._h, _ptr(out), rows, width))
        return out

    def map(self, fn):
        """``[fn(rank, context) for ...]`` with one Python thread per device (ctypes releases the GIL inside library calls); the
        first exception is re-raised."""
        world = len(self.contexts)
        if world == 1:
            return [fn(0, self.contexts[0])]
        out, errs = [None] * world, [None] * world

        def run(r):
            try:
                out[r] = fn(r, self.contexts[r])
            except BaseException as exc:        # noqa: BLE001 -- re-raised below on the calling thread
                errs[r] = exc
        threads = [threading.Thread(target=run, args=(r,), name=f"gsum-dev{self.devices[r]}") for r in range(1, world)]
        for t in threads:
            t.start()
        run(0)
        for t in threads:
            t.join()
        for e in errs:
            if e is not None:
                raise e
        return out


_default_ctx = {}
_lab_ctx = {}
_groups = {}


def device_count() -> int:
    """GPUs the HIP runtime shows this process (0 without one); does not create a context."""
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            hip = C.CDLL(name)
        except OSError:
            continue
        cnt = C.c_int(0)
        return int(cnt.value) if hip.hipGetDeviceCount(C.byref(cnt)) == 0 else 0
    return 0


def resolve_devices(devices):
    """``devices=`` argument of the model classes -> list of device indices ("all": every visible GPU)."""
    if isinstance(devices, str):
        if devices != "all":
            raise ValueError('devices must be "all", an int or a sequence of device indices')
        n = device_count()
        if n < 1:
            raise RuntimeError("devices='all': no HIP device visible (the 'hip' backend has no CPU fallback)")
        return list(range(n))
    if isinstance(devices, (int, np.integer)):
        return [int(devices)]
    ids = [int(d) for d in devices]
    if not ids:
        raise ValueError("devices must not be empty")
    return ids


def default_group(devices="all", lab: bool = False) -> HipGroup:
    """Process-wide group over ``devices`` (adopting ``default_context`` of each)."""
    ids = tuple(resolve_devices(devices))
    key = (ids, bool(lab))
    g = _groups.get(key)
    if g is None or g._h is None or any(c._h is None for c in g.contexts):
        g = HipGroup(list(ids), lab=lab)
        _groups[key] = g
    return g


def _close_default_contexts():
    # Streams with a CU mask must be gone before the C++ finalisers of the HIP runtime / a profiler run (rocprofv3
    # segfaults in __cxa_finalize on a process that exits with one alive): destroy the contexts first.  DeviceMatrix
    # objects still alive only lose their handle (their frees become no-ops).
    for grp in list(_groups.values()) + list(_live_groups):        # groups first: they borrow the contexts below
        try:
            grp.close()
        except Exception:
            pass
    _groups.clear()
    for ctx in list(_default_ctx.values()) + list(_lab_ctx.values()) + list(_live_contexts):
        try:
            ctx.close()
        except Exception:
            pass
    _default_ctx.clear()
    _lab_ctx.clear()


import atexit  # noqa: E402

atexit.register(_close_default_contexts)


def default_context(device: int | None = None) -> HipContext:
    """Process-wide context for ``device`` (default: $LOCAL_RANK or 0)."""
    if device is None:
        device = int(os.environ.get("GSUM_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    ctx = _default_ctx.get(device)
    if ctx is None or ctx._h is None:
        ctx = HipContext(device)
        _default_ctx[device] = ctx
    return ctx


def lab_context(device: int | None = None) -> HipContext:
    """Process-wide context on the LAB build of the library (libgsum_hip_lab.so: include/gsum_hip_debug.h's diagnostics, probes and
    schedule switches on top of the product ABI).  A context of its own -- streams, workspaces -- beside ``default_context``'s."""
    if device is None:
        device = int(os.environ.get("GSUM_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    ctx = _lab_ctx.get(device)
    if ctx is None or ctx._h is None:
        ctx = HipContext(device, lab=True)
        _lab_ctx[device] = ctx
    return ctx
