"""TruncationGP on MI355X — host-side mirror of gsum/models.py:1285-1516.

The truncation layer is thin bookkeeping around the conjugate GP: convert partial sums to
coefficients, hand them to the coefficient process, subtract the Jacobian term.  The likelihood
grid scan that the reference spells as a nested Python loop over ``log_marginal_likelihood``
(docs/notebooks/correlated_EFT_publication.ipynb:1457-1459) is ``log_marginal_likelihood_grid``.
"""
from __future__ import annotations

import warnings

import numpy as np

from ._lib import GSUM_MAX_RHS
from .conjugate import ConjugateGaussianProcess, ConjugateStudentProcess
from .kernels import describe_thetas
from .series import coefficients, geometric_sum

__all__ = ["TruncationGP", "TruncationTP"]


class TruncationGP:
    """Same constructor and methods as gsum.TruncationGP (models.py:1307-1335, 1510-1516)."""

    _coeffs_process_class = ConjugateGaussianProcess

    def __init__(self, kernel=None, ratio=0.5, ref=1, excluded=None, ratio_kws=None, **kwargs):
        # ref / ratio: a number (broadcast over the points) or a callable of X (models.py:1309-1317); kept under the reference's
        # attribute names because user code calls gp.ratio(X, **kws) / gp.ref(X)
        self.ref = self._per_point("ref", ref)
        self.ratio = self._per_point("ratio", ratio)
        self.kernel, self.excluded = kernel, excluded
        self.ratio_kws = {} if ratio_kws is None else ratio_kws          # the caller's dict, like models.py:1326
        self.coeffs_process = self._coeffs_process_class(kernel=kernel, **kwargs)
        self.X_train_ = self.y_train_ = self.orders_ = self.dX_ = self.dy_ = self.coeffs_ = None
        self._fit, self._log_like = False, None
        # predict() conditions on cov(Xc, Xc), which carries no nugget (reference quirk Q7).  When that matrix is singular
        # to working precision the device Cholesky fails; by default the conditioning is then retried with the smallest
        # relative diagonal jitter that makes it factorise, with a RuntimeWarning.  True: raise LinAlgError instead.
        self.strict_conditioning = False

    @staticmethod
    def _per_point(name, value):
        """A number becomes a function of X that broadcasts it over the points and can be overridden per call through the keyword
        of the same name -- ``gp.log_marginal_likelihood(theta, ratio=0.45)`` relies on that (models.py:1310, 1315, 1493)."""
        if callable(value):
            return value

        def constant(X, *positional, **override):           # the reference: lambda X, ratio=ratio: ratio * np.ones(X.shape[0])
            extra = set(override) - {name}
            if extra:
                raise TypeError(f"{name}() got an unexpected keyword argument {sorted(extra)[0]!r}")
            if len(positional) > 1 or (positional and name in override):
                raise TypeError(f"{name}() takes the points and at most one value")
            v = positional[0] if positional else override.get(name, value)
            return v * np.ones(np.shape(X)[0])
        return constant

    def _series(self, X, start, end, factor=1.0):
        """(SeriesScale, ref(X), ratio(X)) of the points X for orders start..end: what the device needs to turn a coefficient
        quantity into a partial-sum quantity -- ref_i ref_j S(ratio_i ratio_j), S the geometric sum with the excluded orders left
        out (helpers.py:149-182)."""
        from ._lib import SeriesScale
        return SeriesScale.make(start, end, self.excluded, factor), self.ref(X), self.ratio(X, **self.ratio_kws)

    # -- scaled mean / cov / basis (models.py:1337-1365) -----------------------------------------
    def _order_sum(self, X, start, end):
        """ref(x) * sum over the orders start..end of ratio(x)^n, one value per point."""
        return self.ref(X) * geometric_sum(x=self.ratio(X, **self.ratio_kws), start=start, end=end, excluded=self.excluded)

    def mean(self, X, start=0, end=np.inf):
        return self._order_sum(X, start, end) * self.coeffs_process.mean(X=X)

    def basis(self, X, start=0, end=np.inf):
        return self._order_sum(X, start, end)[:, None] * self.coeffs_process.basis(X)

    def cov(self, X, Xp=None, start=0, end=np.inf):
        """cov_ij = ref_i ref_j S(ratio_i ratio_j) x [coefficient covariance]_ij.  The coefficient covariance is built AND scaled on
        the device (gsum_kernel_build_series) and crosses PCIe once: no n x m host temporaries (the reference forms four,
        models.py:1343-1348; 8 GiB of host traffic per call at n = 16384).  Xp=None is the one-argument kernel call -- WhiteKernel
        noise on the diagonal -- exactly as in the reference, where Xp is replaced by X only after that call (:1344)."""
        gp = self.coeffs_process
        X = np.asarray(X, dtype=float)
        factor, desc = gp._cov_parts(X.shape[1])
        sc, ref_x, ratio_x = self._series(X, start, end, factor)
        if Xp is None:
            return gp._context().kernel_matrix(desc, X, series=(sc, ref_x, ratio_x))
        Xp = np.asarray(Xp, dtype=float)
        return gp._context().kernel_matrix(desc, X, Xp, series=(sc, ref_x, ratio_x, self.ref(Xp), self.ratio(Xp, **self.ratio_kws)))

    def _cov_diag(self, X, start=0, end=np.inf, two_arg=False):
        """``np.diag(self.cov(X[, X], start, end))`` in O(m) on the host: every stationary leaf is exactly 1 at zero distance, so the
        kernel's diagonal is one number -- with the WhiteKernel terms in for the one-argument call (``cov(X)``, models.py:1344 before
        Xp is replaced) and without them when both arguments are given (``cov(X, X)``: models.py:1443, 1466) -- times the reference's
        own elementwise factors ``ref_i^2 S(ratio_i^2)`` (:1345-1348).  The m x m matrix (2 GiB at m = 16384) is never formed."""
        gp = self.coeffs_process
        X = np.asarray(X, dtype=float)
        factor, desc = gp._cov_parts(X.shape[1])
        kd = (desc.without_white() if two_arg else desc).one_arg_diagonal(X)
        ref, ratio = self.ref(X), self.ratio(X, **self.ratio_kws)
        return (ref * ref) * geometric_sum(x=ratio * ratio, start=start, end=end, excluded=self.excluded) * (factor * kd)

    def underlying_properties(self, X, order, return_std=False, return_cov=False):
        """Truncation error beyond ``order``: prior mean [and covariance / its diagonal] of the orders order + 1 .. inf."""
        y_mean = self.mean(X, start=order + 1)
        if not (return_cov or return_std):
            return y_mean
        if return_cov:
            return y_mean, self.cov(X, start=order + 1)
        return y_mean, np.sqrt(self._cov_diag(X, start=order + 1))

    # -- fit (models.py:1367-1387) -------------------------------------------------------------------
    def fit(self, X, y, orders, dX=None, dy=None):
        for name, fn in (("ratio", lambda: self.ratio(X, **self.ratio_kws)), ("ref", lambda: self.ref(X))):
            if np.atleast_1d(fn()).ndim > 1:
                raise ValueError(f'{name} must return a 1d array or a scalar')            # models.py:1379-1382
        self.X_train_, self.y_train_, self.orders_, self.dX_, self.dy_ = X, y, orders, dX, dy
        self.coeffs_, _ = self._coeffs_and_jacobian(X, y, orders, self.ratio_kws)
        self.coeffs_process.fit(X=X, y=self.coeffs_)
        self._fit = True
        return self

    # -- predict (models.py:1389-1483) ----------------------------------------------------------------
    def _condition(self, X, Xc, resid, start, end, want_cov, ctx=None):
        """The conditioning algebra of models.py:1443-1452 / 1464-1473 for K = cov(., ., start, end):
        returns (K_no K_oo^-1 resid, diag(K_no K_oo^-1 K_on), K_no K_oo^-1 K_on or None).

        The reference solves with LU (numpy.linalg.solve) on cov(Xc, Xc), which carries neither nugget nor
        WhiteKernel noise (two-argument kernel call).  Here K_oo is built and scaled on the
        device and factorised by the same Cholesky as the likelihood path: for a positive definite K_oo the two
        agree to rounding x cond(K_oo); a K_oo that is singular to working precision is retried with a small relative
        jitter (RuntimeWarning), or raises LinAlgError when ``strict_conditioning`` is set."""
        from ._lib import SeriesScale
        gp = self.coeffs_process
        ctx = gp._context() if ctx is None else ctx
        X = np.asarray(X, dtype=float)
        Xc = np.asarray(Xc, dtype=float)
        factor, desc = gp._cov_terms(Xc.shape[1])      # coefficient covariance = factor * kernel_desc(X, Xp)
        sc = SeriesScale.make(start, end, self.excluded, factor)
        ref_c, ratio_c = self.ref(Xc), self.ratio(Xc, **self.ratio_kws)
        ref_n, ratio_n = self.ref(X), self.ratio(X, **self.ratio_kws)
        # The reference solves with LU on a matrix that is, for the dense RBF training sets it is used on, singular to
        # working precision (cond >= 1e16) and still returns usable numbers.  A Cholesky factorisation of the same
        # matrix fails; rather than regress those workflows the conditioning is retried with a relative jitter on the
        # diagonal of the correlation matrix, smallest first, and says so.
        # (relative to the kernel's own diagonal: amplitude + additive constant of the descriptor)
        kdiag = float(np.mean(desc.one_arg_diagonal(Xc)))      # (desc carries no white noise here: see _cov_terms; a DotProduct leaf: the mean over the points)
        for jitter in (0.0, 1e-14, 1e-12, 1e-10, 1e-8, 1e-6):
            K, info = ctx.factorize(desc, Xc, diag_add=jitter * kdiag, series=(sc, ref_c, ratio_c))
            try:
                if info == 0:
                    colsumsq, shift, red = ctx.predict_terms(K, desc, Xc, X, rhs=resid, want_cov=want_cov,
                                                             series=(sc, ref_c, ratio_c, ref_n, ratio_n))
                    break
            finally:
                K.free()
            if self.strict_conditioning:
                raise np.linalg.LinAlgError(
                    'cov(Xc, Xc) is not positive definite to working precision (leading minor %d); the reference '
                    'conditions on it with LU and no jitter -- use fewer / better separated conditioning points' % info)
        else:
            raise np.linalg.LinAlgError('cov(Xc, Xc) is not positive definite even with a relative jitter of 1e-6')
        self.conditioning_jitter_ = jitter          # what the last conditioning needed (0.0: none), for callers to check
        if jitter > 0.0:
            warnings.warn('cov(Xc, Xc) is singular to working precision: conditioned with a relative diagonal jitter of %g '
                          '(the reference solves this system with LU and no jitter; set strict_conditioning=True to get '
                          'LinAlgError instead)' % jitter, RuntimeWarning, stacklevel=3)
        return shift[:, 0], colsumsq, red

    def predict(self, X, order, return_std=False, return_cov=False, Xc=None, y=None, pred_noise=False, kind='both', devices=None):
        """``devices`` (additive; "all" or a list of GPU indices): the new points in one block per device, every device conditioning
        on its own copy of ``cov(Xc, Xc)``; mean and standard deviation equal the one-device call (``return_cov`` is refused)."""
        if devices is not None and self._fit:
            if return_cov:
                raise ValueError("return_cov needs every new point on one device: call predict without devices=")
            from .grid import shard_range
            grp = self.coeffs_process._group(devices)
            world = len(grp)
            Xa = np.asarray(X)

            def block(r, ctx):
                lo, hi = shard_range(Xa.shape[0], r, world)
                if hi == lo:
                    return None
                return self._predict_on(Xa[lo:hi], order, return_std, False, Xc, y, pred_noise, kind, ctx)
            parts = [p for p in grp.map(block) if p is not None]
            if return_std:
                return np.concatenate([np.atleast_1d(p[0]) for p in parts]), np.concatenate([np.atleast_1d(p[1]) for p in parts])
            return np.concatenate([np.atleast_1d(p) for p in parts])
        return self._predict_on(X, order, return_std, return_cov, Xc, y, pred_noise, kind, None)

    def _predict_on(self, X, order, return_std, return_cov, Xc, y, pred_noise, kind, ctx):
        if not self._fit:
            return self.underlying_properties(X, order, return_cov=return_cov, return_std=return_std)
        if Xc is None:
            Xc = self.X_train_
        if y is None:
            if order not in self.orders_:
                raise ValueError('order must be in orders passed to `fit`')            # models.py:1423-1424
            if self.y_train_.ndim == 1:
                y = self.y_train_
            else:
                y = np.squeeze(self.y_train_[:, self.orders_ == order])               # models.py:1428
        if kind not in ['both', 'interp', 'trunc']:
            raise ValueError('kind must be one of "both", "interp" or "trunc"')     # models.py:1430-1431
        want_var = return_std or return_cov
        m_pred, K_pred = 0, 0
        if kind == 'both' or kind == 'interp':
            # interpolating prediction of y_order, conditioned on (Xc, y)             models.py:1434-1453
            m_old = self.mean(X=Xc, start=0, end=order)
            m_new = self.mean(X=X, start=0, end=order)
            shift, red_diag, red = self._condition(X, Xc, np.asarray(y, dtype=float) - m_old, 0, order, return_cov, ctx)
            m_pred = m_pred + m_new + shift
            if return_cov:
                K_pred = K_pred + (self.cov(start=0, end=order, X=X, Xp=X) - red)
            elif return_std:                      # the diagonal alone, O(m): no m x m matrix is built or moved
                K_pred = K_pred + (self._cov_diag(X, 0, order, two_arg=True) - red_diag)
        if kind == 'both' or kind == 'trunc':
            # truncation error                                                         models.py:1455-1476
            m_new_trunc = self.mean(X=X, start=order + 1, end=np.inf)
            K_nn_trunc = self.cov(X=X, Xp=X, start=order + 1, end=np.inf) if return_cov else None
            d_nn_trunc = self._cov_diag(X, order + 1, np.inf, two_arg=True) if return_std else None
            if self.dX_ is not None:                                                   # constrained
                m_old_trunc = self.mean(X=self.dX_, start=order + 1, end=np.inf)
                shift, red_diag, red = self._condition(X, self.dX_, np.asarray(self.dy_, dtype=float) - m_old_trunc,
                                                       order + 1, np.inf, return_cov, ctx)
                m_pred = m_pred + m_new_trunc + shift
                if want_var:
                    K_pred = K_pred + (K_nn_trunc - red if return_cov else d_nn_trunc - red_diag)
            else:
                m_pred = m_pred + m_new_trunc
                if want_var:
                    K_pred = K_pred + (K_nn_trunc if return_cov else d_nn_trunc)
        if return_cov:
            return m_pred, K_pred
        if return_std:
            return m_pred, np.sqrt(K_pred)          # K_pred holds the diagonal here     models.py:1482
        return m_pred

    # -- likelihood (models.py:1485-1507) ---------------------------------------------------------------
    def _coeffs_and_jacobian(self, X, y, orders, ratio_kws, want_coeffs=True):
        ref = self.ref(X)
        ratio = self.ratio(X, **ratio_kws)
        orders = np.asarray(orders)
        orders_mask = ~np.isin(orders, self.excluded)                              # models.py:1495
        # (want_coeffs = False: a grid row of mode "reuse" that rescales another row's Gram matrix needs the Jacobian term only)
        coeffs = coefficients(y=y, ratio=ratio, ref=ref, orders=orders)[:, orders_mask] if want_coeffs else None
        orders_in = orders[orders_mask]
        n = len(orders_in)
        det_factor = np.sum(n * np.log(np.abs(ref)) + np.sum(orders_in) * np.log(np.abs(ratio)))   # :1505
        return coeffs, det_factor

    def _rhs_rows(self, X, y, orders, kws_rows):
        """``_coeffs_and_jacobian`` + ``_rhs`` for several ratio settings at once: ([coefficients | 1] as (rows, n, r + 1), Jacobian terms (rows,)).
        The same elementwise expressions as helpers.py:71-101 / models.py:1495-1505 over a leading axis -- the notebook's 80 ratio rows cost 80 x a
        dozen small numpy calls before (2.3 of the grid's 4.4 ms)."""
        n = np.shape(X)[0]
        ref = np.broadcast_to(np.atleast_1d(self.ref(X)), (n,))
        ratios = [np.asarray(self.ratio(X, **kws)) for kws in kws_rows]
        ratios = np.stack([r if r.shape == (n,) else np.broadcast_to(np.atleast_1d(r), (n,)) for r in ratios])
        orders = np.asarray(orders)
        y = np.asarray(y)
        if y.ndim != 2:
            raise ValueError("y must be 2d")
        if len(orders) != y.shape[-1]:
            raise ValueError("partials and orders must have the same length")
        mask = ~np.isin(orders, self.excluded)
        c = np.empty(y.shape, dtype=np.result_type(y, float))
        c[..., 0] = y[..., 0]
        c[..., 1:] = np.diff(y, axis=-1)
        scale = ref[None, :, None] * ratios[:, :, None] ** orders
        Z = np.ones((len(kws_rows), n, int(mask.sum()) + 1))
        Z[:, :, :-1] = (c[None] / scale)[:, :, mask]
        orders_in = orders[mask]
        dets = np.sum(len(orders_in) * np.log(np.abs(ref))[None, :] + np.sum(orders_in) * np.log(np.abs(ratios)), axis=1)
        return Z, dets

    def log_marginal_likelihood(self, theta, eval_gradient=False, X=None, y=None, orders=None, **ratio_kws):
        X = self.X_train_ if X is None else X
        y = self.y_train_ if y is None else y
        orders = self.orders_ if orders is None else orders
        coeffs, det_factor = self._coeffs_and_jacobian(X, y, orders, ratio_kws)
        result = self.coeffs_process.log_marginal_likelihood(theta, eval_gradient=eval_gradient, X=X, y=coeffs)
        # like the reference (:1498-1507) only the value is returned, even when a gradient was requested
        coeff_log_like = result[0] if eval_gradient else result
        return coeff_log_like - det_factor

    def log_marginal_likelihood_grid(self, thetas, ratio_kws_list, scales=None, X=None, y=None, orders=None, mode="full",
                                     shard=None, devices=None, gather="host", _ctx=None):
        """Likelihood surface over (ratio settings) x (thetas) [x (prior scales)].

        ``scales=None``: ``out[i, j]`` equals ``self.log_marginal_likelihood(thetas[j], **ratio_kws_list[i])``.
        ``scales=[cbar_0, ...]``: ``out[i, j, s]`` is the same call on a process whose prior is ``sd=scales[s]``
        (``df0 = inf, scale0 = cbar``: models.py:115-117, 419-422 -- the only way the reference makes cbar a free
        axis), every other constructor argument unchanged: the (cbar, ratio) scan of BASELINE config 4 is
        ``log_marginal_likelihood_grid([theta], ratios, scales=cbars)[:, 0, :]``.

        mode="full"   every grid point runs kernel build + Cholesky + solve, like the reference's
                      nested loop (notebook :1457-1459) -- the throughput-comparable mode;
        mode="reuse"  one factorisation per theta: all ratio settings share it and only the k-column
                      forward solve is repeated; a prior scale only enters the O(k^2) host algebra
                      (never mixed into "full" throughput numbers).
        ``shard=(rank, world)`` evaluates only this rank's share of the grid and leaves the rest NaN:
        a contiguous block of the flattened grid in mode "full", a block of whole thetas in mode
        "reuse" (``gsum_amd.grid.owned_points``; ``lml_grid_distributed(..., partition=...)`` or
        :meth:`log_marginal_likelihood_grid_distributed` gathers accordingly).
        ``devices="all"`` / ``devices=[0, 1, ...]`` uses several GPUs from THIS process (the reference's caller is one process:
        notebook :1444-1459): the same shards, one per device, each on a host thread of its own with that device's context
        (``gsum_amd.HipGroup``), merged into one surface that equals the one-device result bit for bit.  ``gather="rccl"``
        additionally exchanges the shards between the devices with one in-place ``ncclAllGather`` (``gsum_group_allgather``;
        every rank's gathered surface is compared with rank 0's) instead of merging on the host only.
        """
        if devices is not None:
            if shard is not None:
                raise ValueError("shard= (one process per GPU) and devices= (one process, several GPUs) exclude each other")
            if gather not in ("host", "rccl"):
                raise ValueError('gather must be "host" or "rccl"')
            if mode != "full" or self._grid_many_curves(orders):
                return self._grid_over_devices(thetas, ratio_kws_list, scales, X, y, orders, mode, devices, gather)
            # mode "full": the surface is ONE call of the library's device group (gsum_group_lml_resident_sets: the flattened points
            # block-partitioned over the devices, one host thread per device inside the library, the optional RCCL gather there too)
            _ctx = _GroupEngine(self.coeffs_process._group(devices), gather)
        X = self.X_train_ if X is None else X
        y = self.y_train_ if y is None else y
        orders = self.orders_ if orders is None else orders
        Xd = np.asarray(X, dtype=float)
        gp = self.coeffs_process
        base = gp._active_kernel()
        ctx = gp._context() if _ctx is None else _ctx
        ni, nj = len(ratio_kws_list), len(thetas)
        if scales is None:
            ns, scale_vals = 1, None
        else:
            scale_vals = np.atleast_1d(np.asarray(scales, dtype=float))
            if scale_vals.ndim != 1 or scale_vals.size == 0:
                raise ValueError('scales must be a non-empty 1d sequence of prior standard deviations')
            ns = scale_vals.size
        out = np.full((ni, nj, ns), np.nan)
        from .grid import owned_points
        # this rank's grid points.  mode="full": a contiguous block of the C-order flattening (every point is a full evaluation,
        # points of one ratio row share their right-hand sides).  mode="reuse": whole THETAS -- all ratio settings and scales of a
        # theta share its one factorisation, so theta is the unit of work ("group by distinct kernel descriptor
        # first"); a flat block would make every rank factorise every theta.
        mine = owned_points(ni, nj, ns, *(shard or (0, 1)), partition="theta" if mode == "reuse" else "flat")

        def shaped(a):
            return a if scales is not None else a[:, :, 0]

        if not len(mine):
            if mode not in ("full", "reuse"):
                raise ValueError('mode must be "full" or "reuse"')
            return shaped(out)
        n_pts = Xd.shape[0]
        prep = {}
        if self._grid_many_curves(orders):
            if mode not in ("full", "reuse"):
                raise ValueError('mode must be "full" or "reuse"')
            return shaped(self._grid_chunked(out, mine, thetas, ratio_kws_list, scale_vals, Xd, y, orders, mode, ctx))

        def rhs_for(i):
            if i not in prep:
                kws = ratio_kws_list[i]
                kws = kws if isinstance(kws, dict) else {"ratio": kws}
                coeffs, det = self._coeffs_and_jacobian(Xd, y, orders, kws)
                prep[i] = (gp._rhs(Xd, coeffs), det)
            return prep[i]

        def det_for(i):
            """The Jacobian term of ratio row i alone (the same expression rhs_for evaluates, without the coefficients)."""
            if i in prep:
                return prep[i][1]
            kws = ratio_kws_list[i]
            kws = kws if isinstance(kws, dict) else {"ratio": kws}
            return self._coeffs_and_jacobian(Xd, y, orders, kws, want_coeffs=False)[1]

        def lml_values(G, sld, svals):
            """Host algebra for a stack of Gram matrices; ``svals[b]`` is the prior scale of entry b (None: the
            process's own prior)."""
            if scale_vals is None:
                return gp._lml_gram_batch(G, sld, n_pts)
            return gp._lml_gram_batch_sd(G, sld, n_pts, svals)

        # one descriptor per theta this rank touches, built in one go and without scikit-learn's per-theta clone
        js = np.unique((np.asarray(mine, dtype=np.int64) // ns) % nj).tolist()
        desc_of = dict(zip(js, describe_thetas(base, [thetas[j] for j in js], Xd.shape[1])))
        desc_for = desc_of.__getitem__

        if mode == "full":
            # group this rank's points by ratio setting: they share the right-hand sides, so X and Z go to the
            # device once per row and the (theta, scale) points of the row run as ONE pipelined batch (several
            # evaluations in flight on the GPU); every point still does its own kernel build + Cholesky + solve
            mine_a = np.asarray(mine, dtype=np.int64)
            i_a, rest = np.divmod(mine_a, nj * ns)
            j_a, s_a = np.divmod(rest, ns)
            # ... and the rows go to the device TOGETHER: every row's right-hand sides resident as a set of their own, every point naming
            # its set (gsum_lml_resident_sets) -- the whole surface is one call whose rounds follow one another on the device, instead of
            # one call per ratio row with an upload, a drained pipeline and the host algebra in between (in chunks of rows whose sets
            # stay under 1 GiB).  The per-point bookkeeping is numpy's (round 5: 8000 Python iterations and the join of 8000 descriptors
            # were most of the notebook grid's 10.6 ms): the ~100 distinct descriptors are gathered into the call's array in one go.
            row_ids = np.unique(i_a)
            k_rhs = int(np.sum(~np.isin(np.asarray(orders), self.excluded))) + 1
            max_rows = max(1, int((1 << 30) // max(1, n_pts * k_rhs * 8)))
            js_a = np.asarray(js, dtype=np.int64)
            uniq = [desc_of[j] for j in js]
            tile = getattr(ctx, "tile_descs", None) or getattr(getattr(ctx, "group", None), "tile_descs", None) or gp._context().tile_descs
            for lo in range(0, len(row_ids), max_rows):
                chunk = row_ids[lo:lo + max_rows]
                sel = np.nonzero((i_a >= chunk[0]) & (i_a <= chunk[-1]))[0]              # (ascending rows: a chunk is a range)
                kws_rows = [ratio_kws_list[int(i)] for i in chunk]
                Zs, dets = self._rhs_rows(Xd, y, orders, [kws if isinstance(kws, dict) else {"ratio": kws} for kws in kws_rows])
                ctx.set_inputs_sets(Xd, Zs)
                set_of = np.searchsorted(chunk, i_a[sel])
                descs = tile(uniq, np.searchsorted(js_a, j_a[sel]))
                dets = dets[set_of]
                G, sld, info = ctx.lml_resident_sets(descs, set_of, gp.nugget)
                svals = None if scale_vals is None else scale_vals[s_a[sel]]
                vals = np.where(info != 0, -np.inf, lml_values(G, sld, svals) - dets)
                out[i_a[sel], j_a[sel], s_a[sel]] = vals
        elif mode == "reuse":
            by_theta = {}
            for flat in mine:
                i, rest = divmod(int(flat), nj * ns)
                j, s = divmod(rest, ns)
                by_theta.setdefault(j, {}).setdefault(i, []).append(s)
            # A constant ratio only rescales the coefficient curves order by order, c_n(q) = c_n(q0) (q0 / q)^n
            # (helpers.py:101-106), so G(q) = D G(q0) D with D = diag((q0 / q)^orders, 1): the
            # whole ratio axis of a theta then costs ONE forward solve.  Position-dependent ratios solve per setting.
            orders_in = np.asarray(orders)[~np.isin(np.asarray(orders), self.excluded)]
            const_ratio = {}

            def ratio_const(i):
                if i not in const_ratio:
                    kws = ratio_kws_list[i]
                    kws = kws if isinstance(kws, dict) else {"ratio": kws}
                    rv = np.atleast_1d(self.ratio(Xd, **kws))
                    const_ratio[i] = float(rv[0]) if np.all(rv == rv[0]) and rv[0] != 0 else None
                return const_ratio[i]

            for j, rows in by_theta.items():
                L, info = ctx.factorize(desc_for(j), Xd, diag_add=gp.nugget)
                try:
                    anchor = None                                   # (ratio, G, sld) of the first constant-ratio row
                    for i, ss in rows.items():
                        if info != 0:
                            out[i, j, ss] = -np.inf
                            continue
                        q = ratio_const(i)
                        if q is not None and anchor is not None:
                            q0, G0, sld = anchor
                            D = np.append((q0 / q) ** orders_in, 1.0)
                            G = D[:, None] * G0 * D[None, :]
                        else:
                            G, sld = ctx.forward_gram(L, rhs_for(i)[0])
                            if q is not None:
                                anchor = (q, G, sld)
                        svals = None if scale_vals is None else scale_vals[ss]
                        Gs = np.broadcast_to(G, (len(ss),) + G.shape)
                        out[i, j, ss] = lml_values(Gs, np.full(len(ss), sld), svals) - det_for(i)
                finally:
                    L.free()
        else:
            raise ValueError('mode must be "full" or "reuse"')
        return shaped(out)


    def _grid_many_curves(self, orders):
        orders = np.asarray(self.orders_ if orders is None else orders)
        return int(np.sum(~np.isin(orders, self.excluded))) + 1 > GSUM_MAX_RHS

    def _grid_chunked(self, out, mine, thetas, ratio_kws_list, scale_vals, Xd, y, orders, mode, ctx):
        """The surface for more curves than one device call takes (conjugate.py: _rhs_chunks, _stand_in): every (ratio, theta) point
        gathers its Gram matrix from one device call per chunk of curves -- a full evaluation each in mode "full", a forward solve
        against the theta's one factor in mode "reuse" -- and the prior scales share it.  Rare (16 or more orders at once); plain."""
        gp = self.coeffs_process
        ni, nj, ns = out.shape
        n_pts = Xd.shape[0]
        base = gp._active_kernel()
        todo = {}
        for flat in mine:
            i, rest = divmod(int(flat), nj * ns)
            j, s = divmod(rest, ns)
            todo.setdefault(j, {}).setdefault(i, []).append(s)
        descs = dict(zip(todo, describe_thetas(base, [thetas[j] for j in todo], Xd.shape[1])))
        prep = {}

        def chunks_for(i):
            if i not in prep:
                kws = ratio_kws_list[i]
                kws = kws if isinstance(kws, dict) else {"ratio": kws}
                coeffs, det = self._coeffs_and_jacobian(Xd, y, orders, kws)
                prep[i] = (coeffs.shape[1], list(gp._rhs_chunks(Xd, coeffs)), det)
            return prep[i]

        for j, rows in todo.items():
            L = None
            if mode == "reuse":
                L, info = ctx.factorize(descs[j], Xd, diag_add=gp.nugget)
            try:
                for i, ss in rows.items():
                    ny, chunks, det = chunks_for(i)
                    blocks, sld, bad = [], 0.0, mode == "reuse" and info != 0
                    for idx, Zc in chunks:
                        if bad:
                            break
                        if mode == "reuse":
                            Gc, sld = ctx.forward_gram(L, Zc)
                        else:
                            Gc, slds, infos = ctx.lml_batch([descs[j]], Xd, Zc, gp.nugget)
                            Gc, sld, bad = Gc[0], slds[0], infos[0] != 0
                        blocks.append((idx, Gc))
                    if bad:
                        out[i, j, ss] = -np.inf
                        continue
                    Gs = np.broadcast_to(gp._stand_in(blocks, ny), (len(ss), ny + 1, ny + 1))
                    slds = np.full(len(ss), sld)
                    if scale_vals is None:
                        out[i, j, ss] = gp._lml_gram_batch(Gs, slds, n_pts) - det
                    else:
                        out[i, j, ss] = gp._lml_gram_batch_sd(Gs, slds, n_pts, scale_vals[ss]) - det
            finally:
                if L is not None:
                    L.free()
        return out

    def _grid_over_devices(self, thetas, ratio_kws_list, scales, X, y, orders, mode, devices, gather):
        if mode not in ("full", "reuse"):
            raise ValueError('mode must be "full" or "reuse"')
        if gather not in ("host", "rccl"):
            raise ValueError('gather must be "host" or "rccl"')
        grp = self.coeffs_process._group(devices)
        world = len(grp)
        parts = grp.map(lambda r, ctx: np.asarray(self.log_marginal_likelihood_grid(
            thetas, ratio_kws_list, scales=scales, X=X, y=y, orders=orders, mode=mode, shard=(r, world), _ctx=ctx), dtype=np.float64))
        from .grid import owned_points
        out = parts[0].copy()
        ns = out.shape[2] if out.ndim == 3 else 1
        flat = out.reshape(-1)
        for r in range(1, world):                # the shards are disjoint and cover the grid (gsum_shard_range)
            mine = owned_points(out.shape[0], out.shape[1], ns, r, world, partition="theta" if mode == "reuse" else "flat")
            flat[mine] = parts[r].reshape(-1)[mine]
        if gather == "rccl" and out.size:
            # the exchange step on the devices: rank r's block of the surface (flat C-order block in mode "full", whole thetas in
            # mode "reuse": the partitions of gsum_amd.grid.owned_points) travels through ITS device's gather buffer
            if mode == "reuse":
                by_theta = np.ascontiguousarray(np.moveaxis(out, 1, 0))
                full = grp.allgather(by_theta.reshape(by_theta.shape[0], -1)).reshape(by_theta.shape)
                out = np.ascontiguousarray(np.moveaxis(full, 0, 1))
            else:
                out = grp.allgather(out.reshape(-1, 1)).reshape(out.shape)
        return out

    def log_marginal_likelihood_grid_distributed(self, thetas, ratio_kws_list, scales=None, mode="full", group=None, **kwargs):
        """``log_marginal_likelihood_grid`` with the grid sharded over the ranks of the initialised ``torch.distributed`` group
        (one process per GPU) and ONE all-gather of the fp64 surface -- the reference's nested loop over grid points
        (docs/notebooks/correlated_EFT_publication.ipynb:1457-1459) as a data-parallel map.  Without a process group it is
        the plain call."""
        import functools
        from .grid import lml_grid_distributed
        fn = functools.partial(self.log_marginal_likelihood_grid, thetas, ratio_kws_list, scales=scales, mode=mode, **kwargs)
        return lml_grid_distributed(fn, len(ratio_kws_list), len(thetas), group=group, partition="theta" if mode == "reuse" else "flat")


class _GroupEngine:
    """The two calls ``log_marginal_likelihood_grid(mode="full")`` makes, on a device group instead of one context."""

    def __init__(self, group, gather):
        self.group, self.gather = group, gather

    def set_inputs_sets(self, X, rhs_sets):
        self.group.set_inputs_sets(X, rhs_sets)

    def lml_resident_sets(self, descs, set_of, nugget):
        return self.group.lml_resident_sets(descs, set_of, nugget, gather=self.gather)


class TruncationTP(TruncationGP):
    """Student-t truncation process; same surface as gsum.TruncationTP (models.py:1519-1570)."""

    _coeffs_process_class = ConjugateStudentProcess

    def predict(self, X, order, return_std=False, return_cov=False, Xc=None, y=None, pred_noise=False, kind='both'):
        # like the reference (:1527-1530), `kind` is not forwarded: the Gaussian part is always kind='both'
        pred = super().predict(X=X, order=order, return_std=return_std, return_cov=return_cov, Xc=Xc, y=y,
                               pred_noise=pred_noise)
        if not return_std and not return_cov:
            return pred
        if Xc is None:
            Xc = self.X_train_
        X = np.asarray(X, dtype=float)
        var, disp = self.coeffs_process.cov_factor_, self.coeffs_process.disp_
        basis_lower = np.zeros((X.shape[0], disp.shape[0]))
        basis_trunc = np.zeros((X.shape[0], disp.shape[0]))
        if kind == 'both' or kind == 'interp':                              # models.py:1543-1550
            old = self.basis(X=Xc, start=0, end=order)
            shift, _, _ = self._condition(X, Xc, old[:, 0], 0, order, False)
            basis_lower = self.basis(X=X, start=0, end=order) - shift[:, None]
        if kind == 'both' or kind == 'trunc':                               # models.py:1552-1562
            if self.dX_ is not None:
                old = self.basis(X=self.dX_, start=order + 1, end=np.inf)
                shift, _, _ = self._condition(X, self.dX_, old[:, 0], order + 1, np.inf, False)
                basis_trunc = self.basis(X=X, start=order + 1, end=np.inf) - shift[:, None]
            else:
                basis_trunc = self.basis(start=order + 1, end=np.inf, X=X)
        mean_cov = var * (basis_lower + basis_trunc) @ disp @ (basis_lower + basis_trunc).T   # models.py:1564
        if return_std:
            mean, std = pred
            return mean, std + np.sqrt(np.diag(mean_cov))
        mean, cov = pred
        return mean, cov + mean_cov
