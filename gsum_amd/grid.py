"""Sharding of the (ratio, theta) likelihood grid over the GPUs of one node.

Grid points are independent (the reference evaluates them in a nested Python loop,
docs/notebooks/correlated_EFT_publication.ipynb:1457-1459), so the flattened grid is block-partitioned
over ranks — one process per GPU — with no data-path collective; the only exchange is one all-gather
of the fp64 log-likelihood slices (RCCL over xGMI when the process group is "nccl", gloo on CPU).
"""
from __future__ import annotations

import numpy as np

__all__ = ["shard_range", "owned_points", "gather_flat", "lml_grid_distributed", "predict_distributed"]


def shard_range(total: int, rank: int = 0, world: int = 1):
    """Contiguous slice [lo, hi) of ``range(total)`` owned by ``rank`` (ceil-sized blocks): the partition the C ABI
    defines (``gsum_shard_range``, include/gsum_hip.h), asked of the library itself so that a C host and this layer can
    never disagree.  (The function is host arithmetic; calling it needs the built library but no GPU.)"""
    import ctypes as C
    from ._lib import load_library
    lo, hi = C.c_int64(0), C.c_int64(0)
    if load_library().gsum_shard_range(int(total), int(rank), int(world), C.byref(lo), C.byref(hi)) != 0:
        raise ValueError("bad rank/world")
    return int(lo.value), int(hi.value)


def owned_points(n_rows: int, n_cols: int, n_scales: int, rank: int = 0, world: int = 1, partition: str = "flat") -> np.ndarray:
    """C-order flat indices of the (rows = ratio settings, cols = thetas, scales) surface that ``rank`` evaluates.

    ``partition="flat"``: the contiguous block ``shard_range(n_rows * n_cols * n_scales)`` -- full-recompute scans, where every
    point is its own kernel build + Cholesky.  ``partition="theta"``: every point of the thetas ``shard_range(n_cols)`` -- factor-reuse
    scans, where a theta's factorisation serves all its ratio settings and scales (group by distinct kernel
    descriptor first), so a world of 8 on a 64 x 64 (ell, ratio) scan factorises 8 matrices per rank, not 64."""
    if partition == "flat":
        lo, hi = shard_range(n_rows * n_cols * n_scales, rank, world)
        return np.arange(lo, hi)
    if partition != "theta":
        raise ValueError('partition must be "flat" or "theta"')
    jlo, jhi = shard_range(n_cols, rank, world)
    i, j, s = np.meshgrid(np.arange(n_rows), np.arange(jlo, jhi), np.arange(n_scales), indexing="ij")
    return ((i * n_cols + j) * n_scales + s).reshape(-1)


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def gather_flat(local: np.ndarray, total: int, group=None) -> np.ndarray:
    """All-gather the per-rank slices produced with :func:`shard_range` into the full flat array."""
    dist = _dist()
    if dist is None:
        if len(local) != total:
            raise ValueError("single process must hold the whole grid")
        return np.asarray(local, dtype=np.float64)
    import torch
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    chunk = -(-total // world)
    lo, hi = shard_range(total, rank, world)
    if len(local) != hi - lo:
        raise ValueError(f"rank {rank} holds {len(local)} values, expected {hi - lo}")
    dev = torch.device("cpu")
    if dist.get_backend(group) == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
    buf = torch.full((chunk,), float("nan"), dtype=torch.float64)
    buf[: hi - lo] = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64))
    buf = buf.to(dev)
    full_t = torch.empty(world * chunk, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(full_t, buf, group=group)          # one collective, one output tensor
    full = full_t.cpu().numpy()
    pieces = [full[r * chunk: r * chunk + (shard_range(total, r, world)[1] - shard_range(total, r, world)[0])]
              for r in range(world)]
    return np.concatenate(pieces)


def lml_grid_distributed(evaluate, n_rows: int | None = None, n_cols: int | None = None, group=None, partition: str = "flat") -> np.ndarray:
    """Run ``evaluate(shard=(rank, world))`` -> array with this rank's entries filled (NaN elsewhere), then gather
    every rank's slice.  ``evaluate`` is typically
    ``functools.partial(TruncationGP.log_marginal_likelihood_grid, gp, thetas, ratios, mode=...)`` -- with or without
    ``scales=``: the surface is gathered in whatever shape ``evaluate`` returns it ((rows, cols), or
    (rows, cols, scales) for BASELINE config 4's (cbar, ratio) scan), flattened in C order, which is the order
    ``log_marginal_likelihood_grid`` shards in.  ``n_rows`` / ``n_cols`` are optional and only checked.
    ``partition="theta"`` (factor-reuse scans, see :func:`owned_points`): every rank owns whole columns of the surface; the gather
    then moves the theta axis to the front and exchanges equal blocks of thetas (still one all-gather) -- pass it whenever
    ``evaluate`` runs ``mode="reuse"``:
    ``lml_grid_distributed(functools.partial(gp.log_marginal_likelihood_grid, thetas, ratios, mode="reuse"), partition="theta")``.
    An evaluated point is finite or -inf, never NaN: a NaN inside the slice this rank is about to send means ``evaluate`` filled a
    different partition than the one gathered, and raises instead of returning a surface with holes.
    """
    dist = _dist()
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist is not None else (0, 1)
    surface = np.asarray(evaluate(shard=(rank, world)), dtype=np.float64)
    if n_rows is not None and n_cols is not None and tuple(surface.shape[:2]) != (n_rows, n_cols):
        raise ValueError(f"evaluate returned a surface of shape {surface.shape}, expected ({n_rows}, {n_cols}[, scales])")
    total = surface.size
    if partition == "theta":
        by_theta = np.ascontiguousarray(np.moveaxis(surface, 1, 0))                 # (cols, rows[, scales])
        width = by_theta[0].size
        jlo, jhi = shard_range(by_theta.shape[0], rank, world)
        _no_holes(by_theta[jlo:jhi], partition, rank)
        if dist is None or world == 1:
            return surface
        full = _gather_rows(by_theta[jlo:jhi].reshape(-1), by_theta.shape[0], width, group).reshape(by_theta.shape)
        return np.ascontiguousarray(np.moveaxis(full, 0, 1))
    if partition != "flat":
        raise ValueError('partition must be "flat" or "theta"')
    lo, hi = shard_range(total, rank, world)
    _no_holes(surface.reshape(-1)[lo:hi], partition, rank)
    return gather_flat(surface.reshape(-1)[lo:hi], total, group).reshape(surface.shape)


def _no_holes(own, partition, rank):
    if np.isnan(own).any():
        raise ValueError(f"rank {rank}: the slice of partition={partition!r} it is about to gather still holds NaN -- evaluate() filled "
                         'another partition (mode="reuse" scans shard whole thetas: pass partition="theta")')


def predict_distributed(predict, Xnew, n_curves, group=None):
    """Predictive mean and standard deviation with the new points sharded over ranks.

    Columns of the predictive covariance are independent per new point (gsum/models.py:836), so every rank
    — holding its own copy of the factor — evaluates ``predict(Xnew[lo:hi], return_std=True)`` for its block
    and the (m, n_curves) means and (m,) standard deviations come back in one all-gather.  ``predict`` is
    typically the bound ``ConjugateGaussianProcess.predict`` of a model fitted identically on every rank.
    """
    dist = _dist()
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist is not None else (0, 1)
    Xnew = np.asarray(Xnew, dtype=np.float64)
    m = Xnew.shape[0]
    lo, hi = shard_range(m, rank, world)
    if hi > lo:
        mean, std = predict(Xnew[lo:hi], return_std=True)
        mean = np.asarray(mean, dtype=np.float64).reshape(hi - lo, n_curves)
        std = np.asarray(std, dtype=np.float64).reshape(hi - lo)
    else:
        mean, std = np.empty((0, n_curves)), np.empty(0)
    if dist is None or world == 1:
        return np.squeeze(mean), std
    packed = np.concatenate([mean, std[:, None]], axis=1).reshape(-1)      # n_curves + 1 values per point
    full = _gather_rows(packed, m, n_curves + 1, group).reshape(m, n_curves + 1)
    return np.squeeze(full[:, :n_curves]), full[:, n_curves]


def _gather_rows(local_flat, total_rows, width, group=None):
    """All-gather row blocks produced with shard_range(total_rows, ...): ``local_flat`` holds this rank's rows,
    row-major, ``width`` values per row."""
    dist = _dist()
    import torch
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    chunk = -(-total_rows // world)
    lo, hi = shard_range(total_rows, rank, world)
    dev = torch.device("cpu")
    if dist.get_backend(group) == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
    buf = torch.full((chunk * width,), float("nan"), dtype=torch.float64)
    buf[: (hi - lo) * width] = torch.from_numpy(np.ascontiguousarray(local_flat, dtype=np.float64))
    buf = buf.to(dev)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    pieces = []
    for r_, o in enumerate(outs):
        l2, h2 = shard_range(total_rows, r_, world)
        pieces.append(o.cpu().numpy()[: (h2 - l2) * width])
    return np.concatenate(pieces)
