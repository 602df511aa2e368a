"""Multi-process path of the likelihood grid: shard -> evaluate -> all-gather, on CPU with gloo
(world_size 2 and 3).  The per-point evaluator here is the CPU oracle (test infrastructure); on the GPU
box the same sharding code wraps the HIP path (tests/test_gpu_parity.py, bench.py)."""
import functools
import os
import socket

import numpy as np
import pytest

from gsum_amd.grid import gather_flat, lml_grid_distributed, shard_range


def test_shard_range_partitions_everything():
    for total in (0, 1, 7, 64, 4096, 8000):
        for world in (1, 2, 3, 8):
            got = [shard_range(total, r, world) for r in range(world)]
            flat = [i for lo, hi in got for i in range(lo, hi)]
            assert flat == list(range(total))
            assert max(hi - lo for lo, hi in got) <= -(-total // world) if total else True
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_shard_range_is_the_c_abi_partition():
    """gsum_amd.grid asks the library (gsum_shard_range); the statement of record is ceil-sized contiguous blocks."""
    import ctypes as C
    from gsum_amd._lib import load_library
    lib = load_library()
    for total in (0, 1, 5, 63, 64, 65, 4096, 4097):
        for world in (1, 2, 3, 7, 8, 100):
            chunk = -(-total // world)
            for rank in range(world):
                lo, hi = C.c_int64(-1), C.c_int64(-1)
                assert lib.gsum_shard_range(total, rank, world, C.byref(lo), C.byref(hi)) == 0
                assert (lo.value, hi.value) == (min(total, rank * chunk), min(total, rank * chunk + chunk)) == shard_range(total, rank, world)
    lo, hi = C.c_int64(0), C.c_int64(0)
    for bad in ((10, -1, 2), (10, 2, 2), (10, 0, 0), (-1, 0, 1)):
        assert lib.gsum_shard_range(*bad, C.byref(lo), C.byref(hi)) != 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ni, nj, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json
        from sklearn.gaussian_process.kernels import RBF, WhiteKernel
        from oracle import gsum_oracle as orc
        from conftest import load_golden
        g = load_golden("notebook_grid.json")
        X, y = np.array(g["X_train"]), np.array(g["y_train"])
        ls = np.array(g["ls_vals"])[::100 // nj][:nj]
        qs = np.array(g["ratio_vals"])[::80 // ni][:ni]
        kern = RBF(0.2) + WhiteKernel(g["nugget"], noise_level_bounds="fixed")

        def evaluate(shard):
            out = np.full((ni, nj), np.nan)
            lo, hi = shard_range(ni * nj, *shard)
            for flat in range(lo, hi):
                i, j = divmod(flat, nj)
                out[i, j] = orc.trunc_lml(kern, np.log([ls[j]]), X, y, np.array(g["orders"]), ratio=qs[i], ref=g["ref"])
            return out

        full = lml_grid_distributed(evaluate, ni, nj)
        want = np.array(g["grid"])[::80 // ni][:ni][:, ::100 // nj][:, :nj]
        ok = bool(np.allclose(full, want, rtol=1e-11)) and not np.isnan(full).any()
        q.put((rank, ok, full.shape))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,ni,nj", [(2, 4, 5), (3, 5, 4)])
def test_grid_gather_gloo(world, ni, nj):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ni, nj, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert all(shape == (ni, nj) for _, _, shape in res)


def _scales_worker(rank, world, port, q):
    """A (rows, cols, scales) surface -- BASELINE config 4's (cbar, ratio) scan returns one -- through the gather helper:
    every rank fills its C-order slice of ni * nj * ns flat points with a known function of the index."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ni, nj, ns = 3, 4, 5

        def evaluate(shard):
            out = np.full((ni, nj, ns), np.nan)
            lo, hi = shard_range(ni * nj * ns, *shard)
            out.reshape(-1)[lo:hi] = np.sqrt(np.arange(lo, hi) + 0.5)
            return out

        full = lml_grid_distributed(evaluate, ni, nj)
        want = np.sqrt(np.arange(ni * nj * ns) + 0.5).reshape(ni, nj, ns)
        bad_shape = False
        try:
            lml_grid_distributed(evaluate, nj, ni)
        except ValueError:
            bad_shape = True
        q.put((rank, bool(np.array_equal(full, want)) and bad_shape, full.shape))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_grid_gather_with_scales_axis_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scales_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert all(shape == (3, 4, 5) for _, _, shape in res)


def _predict_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sklearn.gaussian_process.kernels import Matern, WhiteKernel
        from oracle import gsum_oracle as orc
        from gsum_amd.grid import predict_distributed
        rng = np.random.RandomState(0)
        X, y = rng.rand(40, 2) * 4, rng.randn(40, 3)
        Xs = rng.rand(11, 2) * 4                       # 11 points over 2 or 3 ranks: ragged last block
        kern = Matern(length_scale=[0.7, 1.3], nu=2.5) + WhiteKernel(1e-6, noise_level_bounds="fixed")
        fit = orc.cgp_fit(kern, X, y)
        mean, std = predict_distributed(lambda Xb, return_std: orc.cgp_predict(fit, Xb, return_std=True), Xs, 3)
        m0, s0 = orc.cgp_predict(fit, Xs, return_std=True)
        q.put((rank, bool(np.allclose(mean, m0, rtol=1e-12) and np.allclose(std, s0, rtol=1e-12)), mean.shape, std.shape))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_predict_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_predict_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    assert all(ms == (11, 3) and ss == (11,) for _, _, ms, ss in res)


def test_gather_flat_single_process():
    v = np.arange(5.0)
    np.testing.assert_array_equal(gather_flat(v, 5), v)
    with pytest.raises(ValueError):
        gather_flat(v, 6)


def _product_grid_worker(rank, world, port, q):
    """The PRODUCT's grid methods (TruncationGP.log_marginal_likelihood_grid[_distributed]) under gloo, on the cpu backend (numpy /
    scipy behind the HIP library's operator interface, gsum_amd/_cpu.py) so that the work each rank does can be counted."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GSUM_BACKEND="cpu")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gsum_amd
        from sklearn.gaussian_process.kernels import RBF, WhiteKernel
        from gsum_amd._cpu import cpu_context
        rng = np.random.RandomState(3)
        n, r = 40, 4
        X = np.sort(rng.rand(n))[:, None] * 6.0
        y = gsum_amd.partials(rng.randn(n, r), ratio=0.5, ref=2.0, orders=np.arange(r))
        gp = gsum_amd.TruncationGP(kernel=RBF(0.3) + WhiteKernel(1e-6, noise_level_bounds="fixed"), ratio=0.5, ref=2.0, center=0, disp=0,
                                   df=1, scale=1, optimizer=None)
        gp.fit(X, y, orders=np.arange(r))
        thetas = [np.log([e]) for e in np.linspace(0.2, 0.5, 8)]
        ratios = list(np.linspace(0.3, 0.7, 6))
        scales = [0.5, 1.0, 2.0]
        ctx = cpu_context()
        out = {}
        for mode in ("reuse", "full"):
            for sc in (None, scales):
                want = gp.log_marginal_likelihood_grid(thetas, ratios, scales=sc, mode=mode)                 # unsharded, on this rank
                before = ctx.calls["potrf"]
                got = gp.log_marginal_likelihood_grid_distributed(thetas, ratios, scales=sc, mode=mode)
                out[(mode, sc is not None)] = (bool(np.array_equal(got, want)) and not np.isnan(got).any(), ctx.calls["potrf"] - before)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_reuse_mode_shards_whole_thetas_gloo(world):
    """"group by distinct kernel descriptor first".  In mode="reuse" a theta's factorisation serves every ratio
    setting and prior scale, so a rank must own whole thetas: rank r of `world` factorises the ~8 / world matrices of its block, not
    all 8 (what a flat block partition of the (ratio, theta, scale) index did in round 3: zero scaling).  mode="full" keeps the flat
    partition, one factorisation per grid point.  Both gather to exactly the unsharded surface."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_product_grid_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_theta, n_ratio, n_scale = 8, 6, 3
    for rank in range(world):
        lo, hi = shard_range(n_theta, rank, world)
        for with_scales in (False, True):
            ok, potrfs = res[rank][("reuse", with_scales)]
            assert ok
            assert potrfs == hi - lo, (rank, potrfs, hi - lo)                      # one factorisation per OWNED theta, whatever the other axes
            ok, potrfs = res[rank][("full", with_scales)]
            flo, fhi = shard_range(n_theta * n_ratio * (n_scale if with_scales else 1), rank, world)
            assert ok and potrfs == fhi - flo
    assert sum(res[r][("reuse", False)][1] for r in range(world)) == n_theta
