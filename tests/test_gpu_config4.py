"""GPU tests added in round 2 (``-m gpu``): BASELINE config 4 -- the (cbar, ratio) likelihood grid -- as a product
path, both at the size the reference itself can answer (golden fixture) and at n = 8192 against the oracle; the
caller-side operator cho_solve; the resident-input and queue-probe behaviour of the C ABI; an initialised RCCL group.
Everything goes through libgsum_hip.so; the oracle is the checker only."""
import os
import warnings

import numpy as np
import pytest
from scipy.linalg import cho_solve

from conftest import load_golden, record_parity

pytestmark = pytest.mark.gpu

import gsum_amd  # noqa: E402
from oracle import gsum_oracle as orc  # noqa: E402  (checker only)
from sklearn.gaussian_process.kernels import RBF, Matern, WhiteKernel, ConstantKernel as C  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    return gsum_amd.default_context(0)


def _gp_drawn(n, r, seed, ell=0.2, dx=0.1, nugget=1e-10):
    X = dx * np.arange(n)[:, None]
    K = RBF(ell)(X)
    K[np.diag_indices_from(K)] += nugget
    c = np.linalg.cholesky(K) @ np.random.RandomState(seed).randn(n, r)
    return X, gsum_amd.partials(c, ratio=0.5, ref=1.0, orders=np.arange(r))


def _plain_report(name, got, want, mag):
    """Plain relative errors of a likelihood surface whose entries are signed sums of terms of magnitude ``mag``: asserted at 1e-10
    where |entry| >= mag / 2, recorded everywhere (how many entries reach 1e-10, the worst one and how strongly it cancels)."""
    rel = np.abs(got - want) / np.abs(want)
    solid = np.abs(want) >= 0.5 * mag
    assert solid.any() and np.all(rel[solid] <= 1e-10), rel[solid].max()
    worst = np.unravel_index(np.argmax(rel), rel.shape)
    record_parity(name, entries=int(got.size), max_rel_to_term_magnitude=float(np.max(np.abs(got - want) / mag)),
                  max_plain_rel_where_entry_ge_half_its_terms=float(rel[solid].max()), entries_ge_half_their_terms=int(solid.sum()),
                  entries_with_plain_rel_le_1em10=int((rel <= 1e-10).sum()), max_plain_rel_all=float(rel.max()),
                  worst_entry_over_its_term_magnitude=float(np.abs(want[worst]) / mag[worst]))


def test_cbar_ratio_grid_golden():
    """log_marginal_likelihood_grid(scales=...) against the reference's own numbers (tests/golden/cbar_ratio_grid.json:
    every entry one TruncationGP(sd=cbar).log_marginal_likelihood(theta, ratio=q) call of the reference), both modes,
    plus the (ell, ratio) strip; argmax indices exact and interior, no -inf on either side."""
    g = load_golden("cbar_ratio_grid.json")
    X, y = _gp_drawn(g["n"], g["r"], g["seed"])
    gp = gsum_amd.TruncationGP(kernel=RBF(0.2), ratio=0.5, ref=1.0, center=0, disp=0, df=1, scale=1, optimizer=None)
    gp.fit(X, y, orders=np.arange(g["r"]))
    want = np.array(g["grid_ratio_by_cbar"])
    theta = np.log([g["length_scale"]])
    # every entry is a signed sum of terms of very different size (quadratic form / cbar^2, log-determinant, constants)
    # that cancel near the maximum: 1e-10 is relative to the terms' magnitude, as for a single log-likelihood
    _, mag = orc.cbar_ratio_grid_one_factor(RBF(0.2), theta, X, y, np.arange(g["r"]), g["ratios"], g["cbars"], return_scale=True)
    for mode in ("full", "reuse"):
        got = gp.log_marginal_likelihood_grid([theta], g["ratios"], scales=g["cbars"], mode=mode)
        assert got.shape == (len(g["ratios"]), 1, len(g["cbars"]))
        assert np.all(np.abs(got[:, 0, :] - want) <= 1e-10 * mag), np.max(np.abs(got[:, 0, :] - want) / mag)
        # the plain relative error |got - want| / |want| as well.  An entry that is the difference of terms
        # 1000 x its own size cannot be better than 1000 x the error of those terms, so: plain 1e-10 is ASSERTED where the entry is
        # at least half the size of its terms, and what is achieved everywhere is recorded (gpurun_out/parity_achieved.json)
        _plain_report(f"cbar_ratio_grid_golden_{mode}", got[:, 0, :], want, mag)
        assert list(np.unravel_index(np.argmax(got[:, 0, :]), want.shape)) == g["argmax"]
        assert not np.isneginf(got).any()
    # the scales axis is the `sd` prior: one column equals a process constructed with sd = cbar
    gp_sd = gsum_amd.TruncationGP(kernel=RBF(0.2), ratio=0.5, ref=1.0, center=0, disp=0, sd=g["cbars"][2], optimizer=None)
    gp_sd.fit(X, y, orders=np.arange(g["r"]))
    col = gp_sd.log_marginal_likelihood_grid([theta], g["ratios"], mode="full")[:, 0]
    assert np.all(np.abs(col - want[:, 2]) <= 1e-10 * mag[:, 2])
    strip = gp.log_marginal_likelihood_grid([np.log([e]) for e in g["ells"]], g["ratios"], mode="full")
    # (ell, ratio) strip with the default prior: no cancellation between huge terms, but cond(R) grows to 1e13 at ell = 0.3
    strip_rel = {}
    for jj, e in enumerate(g["ells"]):
        K = RBF(e)(X) + 1e-10 * np.eye(len(X))
        cond = float(np.linalg.cond(K))
        wantj = np.array(g["strip_ratio_by_ell"])[:, jj]
        np.testing.assert_allclose(strip[:, jj], wantj, rtol=max(1e-10, 1e-15 * cond))
        strip_rel[f"ell={e:.3f}"] = dict(cond=cond, max_plain_rel=float(np.max(np.abs(strip[:, jj] - wantj) / np.abs(wantj))))
        if cond <= 1e9:                          # the conditioned columns: the north star's plain 1e-10
            assert strip_rel[f"ell={e:.3f}"]["max_plain_rel"] <= 1e-10
    record_parity("ell_ratio_strip_golden", **strip_rel)
    assert list(np.unravel_index(np.argmax(strip), strip.shape)) == g["strip_argmax"]
    # sharded evaluation fills exactly this rank's slice of the flattened (ratio, theta, cbar) grid
    part = gp.log_marginal_likelihood_grid([theta], g["ratios"], scales=g["cbars"], mode="full", shard=(1, 3))
    lo, hi = gsum_amd.shard_range(want.size, 1, 3)
    flat = part.reshape(-1)
    assert np.isnan(flat[:lo]).all() and np.isnan(flat[hi:]).all()
    assert np.all(np.abs(flat[lo:hi] - want.reshape(-1)[lo:hi]) <= 1e-10 * mag.reshape(-1)[lo:hi])
    with pytest.raises(ValueError):
        gp.log_marginal_likelihood_grid([theta], g["ratios"], scales=[])


@pytest.fixture(scope="module")
def config4():
    """S3 inputs with GP-drawn coefficients (so that the likelihood maximum is interior), the fitted process and the
    64 x 64 (cbar, ratio) grid of BASELINE config 4 in factor-reuse mode."""
    n, r = 8192, 6
    X, y = _gp_drawn(n, r, seed=1)
    orders = np.arange(r)
    cbars = np.geomspace(0.25, 4, 64)
    ratios = np.linspace(0.3, 0.7, 64)
    theta = np.log([0.2])
    gp = gsum_amd.TruncationGP(kernel=RBF(0.2), ratio=0.5, ref=1.0, center=0, disp=0, df=1, scale=1, optimizer=None)
    gp.fit(X, y, orders=orders)
    got = gp.log_marginal_likelihood_grid([theta], list(ratios), scales=cbars, mode="reuse")[:, 0, :]
    # the oracle's one-factorisation checker (one LAPACK Cholesky + the reference's solves per ratio) and, per entry, the
    # magnitude of the terms it is the signed sum of (they cancel near the maximum: tolerances are relative to them)
    want, mag = orc.cbar_ratio_grid_one_factor(RBF(0.2), theta, X, y, orders, ratios, cbars, return_scale=True)
    return dict(X=X, y=y, orders=orders, cbars=cbars, ratios=ratios, theta=theta, gp=gp, reuse=got, want=want, mag=mag)


def test_config4_grid_n8192_all_entries_vs_oracle(config4):
    """BASELINE config 4 at full size: n = 8192, 64 x 64 (cbar in geomspace(0.25, 4), ratio in linspace(0.3, 0.7)),
    6 orders.  All 4096 entries against the oracle's one-factorisation checker (one LAPACK Cholesky + the reference's
    solves per ratio; pinned to the reference at n = 192 by tests/test_oracle_golden.py); argmax indices equal and
    interior, -inf count equal (zero)."""
    c = config4
    want, mag, got = c["want"], c["mag"], c["reuse"]
    assert got.shape == (64, 64)
    assert np.isneginf(got).sum() == np.isneginf(want).sum() == 0
    # 1e-10 relative to the magnitude of the terms each entry is the signed sum of (they cancel near the maximum)
    assert np.all(np.abs(got - want) <= 1e-10 * mag), np.max(np.abs(got - want) / mag)
    # ... and in plain relative terms wherever the entry is not a near-total cancellation of its terms
    _plain_report("config4_64x64_n8192_reuse_vs_one_factor_oracle", got, want, mag)
    am_got = np.unravel_index(np.argmax(got), got.shape)
    assert am_got == np.unravel_index(np.argmax(want), want.shape)
    assert 0 < am_got[0] < 63 and 0 < am_got[1] < 63, am_got
    assert abs(c["ratios"][am_got[0]] - 0.5) < 0.02 and abs(np.log(c["cbars"][am_got[1]])) < 0.1   # the generating values


def test_config4_grid_n8192_full_recompute_subgrid(config4):
    """mode="full" -- every point its own K build + Cholesky + solve, the throughput-comparable mode -- on a 3 x 5
    sub-grid (15 whole evaluations through the batch pipeline) against the reuse-mode values, and the sharded call."""
    c = config4
    ri, ci = [0, 31, 63], [0, 20, 32, 45, 63]
    full = c["gp"].log_marginal_likelihood_grid([c["theta"]], list(c["ratios"][ri]), scales=c["cbars"][ci], mode="full")[:, 0, :]
    assert np.all(np.abs(full - c["reuse"][np.ix_(ri, ci)]) <= 1e-10 * c["mag"][np.ix_(ri, ci)])
    part = c["gp"].log_marginal_likelihood_grid([c["theta"]], list(c["ratios"][ri]), scales=c["cbars"][ci], mode="full",
                                                shard=(7, 8))        # rank 7 of 8: the last two points
    flat = part.reshape(-1)
    lo, hi = gsum_amd.shard_range(15, 7, 8)
    assert np.isnan(flat[:lo]).all() and not np.isnan(flat[lo:hi]).any()
    np.testing.assert_array_equal(flat[lo:hi], full.reshape(-1)[lo:hi])


@pytest.mark.parametrize("pts", [[(0, 0), (63, 63), (31, 32), (10, 50)], [(50, 10), (0, 63), (63, 0), None]])
def test_config4_grid_n8192_points_vs_full_oracle(config4, pts):
    """An 8-point subsample in the reference's own spelling: one full CPU evaluation per point,
    orc.trunc_lml(..., sd=cbar) = TruncationGP(sd=cbar).log_marginal_likelihood(theta, ratio=q) (None = the argmax)."""
    c = config4
    got = c["reuse"]
    for pt in pts:
        a, b = pt if pt is not None else np.unravel_index(np.argmax(got), got.shape)
        ref = orc.trunc_lml(RBF(0.2), c["theta"], c["X"], c["y"], c["orders"], ratio=c["ratios"][a], ref=1.0, center=0,
                            disp=0, sd=c["cbars"][b])
        assert abs(got[a, b] - ref) <= 1e-10 * c["mag"][a, b], (a, b)
        if abs(ref) >= 0.5 * c["mag"][a, b]:
            assert abs(got[a, b] - ref) <= 1e-10 * abs(ref), (a, b, abs(got[a, b] - ref) / abs(ref))
        record_parity(f"config4_point_{a}_{b}_vs_full_oracle", plain_rel=float(abs(got[a, b] - ref) / abs(ref)),
                      rel_to_term_magnitude=float(abs(got[a, b] - ref) / c["mag"][a, b]))


def test_ell_ratio_strip_n8192_full_recompute_vs_oracle(config4):
    """The reference-faithful (ell, ratio) scan of SURVEY.md 8(d) at n = 8192 in full-recompute mode: a 64-point strip
    (16 length scales x 4 ratios; every column a distinct factorisation), two points against the oracle, all of them
    against mode="reuse" (whose ratio axis comes from the rescaling identity); argmax at the generating values."""
    c = config4
    gp = c["gp"]
    ells = np.linspace(0.17, 0.21, 16)           # beyond ell ~ 0.2 the matrix at dx = 0.1 is too ill-conditioned for 1e-10
    ratios = [0.4, 0.5, 0.55, 0.6]
    thetas = [np.log([e]) for e in ells]
    full = gp.log_marginal_likelihood_grid(thetas, ratios, mode="full")
    reuse = gp.log_marginal_likelihood_grid(thetas, ratios, mode="reuse")
    assert full.shape == (4, 16) and np.isfinite(full).all()
    np.testing.assert_allclose(full, reuse, rtol=1e-10)
    record_parity("ell_ratio_strip_n8192_full_vs_reuse", max_plain_rel=float(np.max(np.abs(full - reuse) / np.abs(reuse))))
    for i, j in ((1, 8), (3, 11)):               # ell = 0.1913 and 0.1993
        want = orc.trunc_lml(RBF(0.2), thetas[j], c["X"], c["y"], c["orders"], ratio=ratios[i], ref=1.0, center=0, disp=0,
                             df=1, scale=1)
        assert full[i, j] == pytest.approx(want, rel=1e-10), (i, j)
    i, j = np.unravel_index(np.argmax(full), full.shape)
    assert ratios[i] == 0.5 and abs(ells[j] - 0.2) < 0.01 and 0 < j < 15


@pytest.mark.parametrize("n,k", [(1, 1), (77, 3), (128, 16), (129, 2), (1000, 7), (2500, 20)])
def test_cho_solve_matches_scipy(ctx, n, k):
    """gsum_cho_solve = scipy.linalg.cho_solve((L, True), B) (models.py:479), both triangular solves on the device."""
    rng = np.random.RandomState(n + k)
    X = np.sort(rng.rand(n, 2), axis=0) * np.array([8.0, 5.0]) * max(1.0, n / 100.0) ** 0.5
    kern = C(1.4) * Matern([0.7, 1.1], nu=2.5) + WhiteKernel(1e-4)
    R = kern(X)
    B = rng.randn(n, k)
    Ld = ctx.kernel_matrix_dev(gsum_amd.describe_kernel(kern, 2), X)
    assert ctx.potrf(Ld) == 0
    got = ctx.cho_solve(Ld, B)
    L = np.linalg.cholesky(R)
    want = cho_solve((L, True), B)
    tol = 1e-13 * np.linalg.cond(R)
    np.testing.assert_allclose(got, want, rtol=0, atol=tol * np.abs(want).max())
    np.testing.assert_allclose(R @ got, B, rtol=0, atol=1e-9 * np.abs(want).max())
    # the reference's classmethod surface accepts the device factor in place of the host one
    via = gsum_amd.ConjugateGaussianProcess.solve_sqrt(Ld, B[:, 0], 'cholesky')
    np.testing.assert_array_equal(via, got[:, 0])
    basis = np.ones((n, 1))
    if k >= 2:
        c_dev = gsum_amd.ConjugateGaussianProcess.compute_center(B[:, :2], Ld, basis, np.array([0.2]), np.array([[1.5]]), 'cholesky')
        c_host = gsum_amd.ConjugateGaussianProcess.compute_center(B[:, :2], L, basis, np.array([0.2]), np.array([[1.5]]), 'cholesky')
        np.testing.assert_allclose(c_dev, c_host, rtol=1e-8)
    Ld.free()


def test_resident_inputs_survive_other_calls(ctx):
    """Only gsum_set_inputs writes the resident inputs (advisor finding, round 1): operator-level calls, lml_batch and
    lml_grad on OTHER data in between must not change what lml_resident evaluates."""
    rng = np.random.RandomState(5)
    n, k = 300, 4
    X = np.sort(rng.rand(n, 1), axis=0) * 30
    Z = np.c_[rng.randn(n, k - 1), np.ones(n)]
    desc = gsum_amd.describe_kernel(RBF(0.7), 1)
    ctx.set_inputs(X, Z)
    assert ctx.resident_shape() == (n, 1, k)
    G0, s0, i0 = ctx.lml_resident([desc], 1e-8)
    # other data, other shapes, through every other entry point
    X2 = rng.rand(517, 2) * 9
    Z2 = rng.randn(517, 7)
    d2 = gsum_amd.describe_kernel(Matern([0.5, 0.9], nu=1.5) + WhiteKernel(1e-3), 2)
    ctx.lml_batch([d2], X2, Z2, 1e-8)
    kern2 = C(1.2) * RBF([0.5, 0.9]) + WhiteKernel(1e-3)
    from gsum_amd.kernels import describe_gradient
    ctx.lml_grad(gsum_amd.describe_kernel(kern2, 2), describe_gradient(kern2, 2), X2, Z2, 1e-8)
    L2 = ctx.kernel_matrix_dev(d2, X2)
    ctx.potrf(L2)
    ctx.forward_gram(L2, Z2)
    ctx.predict_terms(L2, d2, X2, X2[:9], rhs=Z2[:, :3])
    L2.free()
    assert ctx.resident_shape() == (n, 1, k)
    G1, s1, i1 = ctx.lml_resident([desc], 1e-8)
    np.testing.assert_array_equal(G0, G1)
    assert s0[0] == s1[0] and i0[0] == i1[0] == 0


def test_lml_resident_shard_fills_exactly_its_slice(ctx):
    """The C ABI's multi-GPU entry: rank r of `world` evaluates descriptors [lo, hi) of the list into THEIR positions of
    full-length arrays; the ranks' slices together are the unsharded call, bit for bit (here the ranks run one after
    the other on the one GPU: the all-gather is the host's)."""
    rng = np.random.RandomState(11)
    n, k = 700, 5
    X = np.sort(rng.rand(n, 1), axis=0) * 50
    Z = np.c_[rng.randn(n, k - 1), np.ones(n)]
    descs = [gsum_amd.describe_kernel(RBF(ell), 1) for ell in np.linspace(0.4, 1.4, 11)]
    ctx.set_inputs(X, Z)
    G0, s0, i0 = ctx.lml_resident(descs, 1e-8)
    for world in (1, 3, 4, 16):
        G = np.full_like(G0, np.nan)
        s = np.full_like(s0, np.nan)
        seen = np.zeros(len(descs), dtype=int)
        for rank in range(world):
            Gr, sr, ir, lo, hi = ctx.lml_resident_shard(descs, 1e-8, rank, world)
            assert (lo, hi) == gsum_amd.shard_range(len(descs), rank, world)
            assert np.all(np.isnan(sr[:lo])) and np.all(np.isnan(sr[hi:])) and np.all(ir[:lo] == -1) and np.all(ir[hi:] == -1)
            assert np.all(ir[lo:hi] == 0)
            G[lo:hi], s[lo:hi] = Gr[lo:hi], sr[lo:hi]
            seen[lo:hi] += 1
        assert np.all(seen == 1)
        np.testing.assert_array_equal(G, G0)
        np.testing.assert_array_equal(s, s0)
    with pytest.raises(ValueError):
        ctx.lml_resident_shard(descs, 1e-8, 3, 3)


def test_batches_run_on_three_streams_whatever_the_hardware_queue_count(ctx):
    """Round 4: a batch advances in groups with ONE launch per kernel class and outer step (gs_lml_wave); it owns a chain
    stream per group and one bulk stream -- 4 with the default three groups, the HIP runtime's default number of hardware
    queues -- and this process never asked for more (tests/conftest.py no longer sets GPU_MAX_HW_QUEUES).  Group layout is
    scheduling only: 1 x 24, 2 x 10, 3 x 5 and 4 x 2 give the same bits, the non-positive-definite member included."""
    assert "GPU_MAX_HW_QUEUES" not in os.environ
    rng = np.random.RandomState(1)
    n = 2304
    X = 0.1 * np.arange(n)[:, None]
    Z = np.c_[rng.randn(n, 3), np.ones(n)]
    descs = [gsum_amd.describe_kernel(RBF(0.2 + 0.001 * i), 1) for i in range(23)] + [gsum_amd.describe_kernel(RBF(40.0), 1)]
    old = {k: ctx.get_option(k) for k in ("wave_groups", "wave_size")}
    ctx.set_option("medium_path", 0)
    try:
        ctx.set_inputs(X, Z)
        ref = None
        for groups, size in ((2, 10), (1, 24), (3, 5), (4, 2)):
            ctx.set_option("wave_groups", groups)
            ctx.set_option("wave_size", size)
            G, sld, info = ctx.lml_resident(descs, 0.0)          # no nugget: RBF(40) on this grid is singular to working precision
            assert np.all(info[:-1] == 0) and info[-1] > 0
            if ref is None:
                ref = (G, sld, info)
                assert ctx.get_option("wave_streams") == 3
            np.testing.assert_array_equal(info, ref[2])
            np.testing.assert_array_equal(G[:-1], ref[0][:-1])
            np.testing.assert_array_equal(sld[:-1], ref[1][:-1])
        # and each member equals its own single evaluation (the look-ahead / persistent-chain schedule of one factorisation)
        for b in (0, 7, 22):
            G1, s1, i1 = ctx.lml_resident([descs[b]], 0.0)
            np.testing.assert_array_equal(G1[0], ref[0][b])
            assert s1[0] == ref[1][b] and i1[0] == 0
    finally:
        ctx.set_option("medium_path", 1)
        for k, v in old.items():
            ctx.set_option(k, v)


def test_potrf_info_pattern_matches_lapack_on_notebook_like_inputs(ctx):
    """S0 stress inputs: X = linspace(0, 1, n), nugget 1e-10 -- the reference's own workload shape,
    pivots of order sqrt(nugget).  Which grid entries are -inf is part of 'bit-exact on indices': over a sweep of length
    scales the success / failure pattern of gsum_potrf_lower must equal numpy.linalg.cholesky's."""
    for n in (512, 2048):
        X = np.linspace(0, 1, n)[:, None]
        ells = np.linspace(0.05, 0.5, 64) if n == 512 else np.linspace(0.05, 0.5, 16)
        descs = [gsum_amd.describe_kernel(RBF(float(e)), 1) for e in ells]
        want = []
        for e in ells:
            K = RBF(float(e))(X)
            K[np.diag_indices_from(K)] += 1e-10
            try:
                np.linalg.cholesky(K)
                want.append(True)
            except np.linalg.LinAlgError:
                want.append(False)
        ctx.set_option("medium_path", 0)
        try:
            _, _, info = ctx.lml_batch(descs, X, np.ones((n, 1)), 1e-10)
        finally:
            ctx.set_option("medium_path", 1)
        got = [int(i) == 0 for i in info]
        assert got == want, (n, [(float(e), g, w) for e, g, w in zip(ells, got, want) if g != w])
        # the one-workgroup-per-evaluation path must agree with the multi-kernel path on the pattern
        _, _, info_m = ctx.lml_batch(descs, X, np.ones((n, 1)), 1e-10)
        assert [int(i) == 0 for i in info_m] == got


def test_grid_gather_under_an_initialised_rccl_group():
    """north_star: 'RCCL gather over xGMI of the log-likelihood grid'.  A world-size-1 group with the nccl backend
    (= RCCL on ROCm) is initialised and lml_grid_distributed runs the HIP evaluator under it: the all-gather goes
    through RCCL, the result equals the unsharded call and the reference's numbers.  Runs in a child process
    (tests/rccl_world1_check.py): a communicator owns hardware queues for the life of its process, and the device
    time-slices user compute queues beyond 24 -- the pytest process keeps its own for the other tests."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_check.py")], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "RCCL_WORLD1_OK backend=nccl" in res.stdout


def test_large_cases_against_extended_precision_truth():
    """How far is the HIP path from the TRUE value of the fp64 inputs' log-likelihood, next to LAPACK's distance from it?
    tests/golden/large_truth.json holds the S2 / S3 values (white-noise and GP-drawn coefficients) evaluated in 80-bit
    long double (tests/golden/make_truth.py, oracle/truth_ld.c) on the bit-identical K and coefficients.  The kernel
    matrix on the device equals scikit-learn's bit for bit (test_kernel_matrix_matches_sklearn), so both paths factorise
    the same matrix; what differs is the order of fp64 operations.  Bound: within 1e-10 relative of the truth in every
    case; the measured errors of both paths are written to gpurun_out/truth_errors.json."""
    import json
    from conftest import ROOT
    cases = load_golden("large_truth.json")["cases"]
    rows = []
    for case in cases:
        n, r, q = case["n"], case["r"], case["ratio"]
        X = 0.1 * np.arange(n)[:, None]
        z = np.random.RandomState(case["seed"]).randn(n, r)
        if case["kind"] == "gp_drawn":
            K = RBF(0.2)(X)
            K[np.diag_indices_from(K)] += 1e-10
            z = np.linalg.cholesky(K) @ z
        y = gsum_amd.partials(z, ratio=0.5, ref=1.0, orders=np.arange(r))
        gp = gsum_amd.TruncationGP(kernel=RBF(0.2), ratio=0.5, ref=1.0, center=0, disp=0, df=1, scale=1, optimizer=None)
        gp.X_train_, gp.y_train_, gp.orders_ = X, y, np.arange(r)
        got = gp.log_marginal_likelihood(theta=np.log([0.2]), ratio=q)
        truth = case["lml_truth_f64"]
        rows.append(dict(n=n, kind=case["kind"], ratio=q, hip=got, truth=truth, lapack=case["lml_lapack"],
                         hip_rel_err=abs(got - truth) / abs(truth), lapack_rel_err=case["lapack_rel_err"],
                         hip_vs_lapack=abs(got - case["lml_lapack"]) / abs(truth)))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "truth_errors.json"), "w") as f:
        json.dump(rows, f, indent=1)
    for row in rows:
        assert row["hip_rel_err"] <= 1e-10, row


def test_predict_vs_oracle_n4096_m1024_eight_curves():
    """BASELINE config 5's path at a size the oracle answers in seconds: n = 4096 2-D points, Matern-5/2 + White(1e-6),
    8 curves, 1024 new points -- predictive mean, standard deviation and the full predictive covariance (the sweep runs
    K = 256 two-column steps, the covariance reduction is a lower-tile SYRK mirrored on the device).  Variance tolerance
    1e-10 * cov_factor_ (two valid fp64 formulations already differ by 5e-11 relative)."""
    n, m, r = 4096, 1024, 8
    rng = np.random.RandomState(3)
    side = np.array([0.35, 0.65]) * np.sqrt(n)
    X = rng.rand(n, 2) * side
    Xs = rng.rand(m, 2) * side
    y = rng.randn(n, r) + 0.3
    kern = Matern(length_scale=[0.7, 1.3], nu=2.5) + WhiteKernel(1e-6, noise_level_bounds="fixed")
    priors = dict(center=0.1, disp=1.5, df=2, scale=0.7)
    gp = gsum_amd.ConjugateGaussianProcess(kernel=kern, optimizer=None, **priors)
    gp.fit(X, y)
    fit = orc.cgp_fit(kern, X, y, **priors)
    assert gp.cov_factor_ == pytest.approx(fit["cov_factor"], rel=1e-10)
    mean, std = gp.predict(Xs, return_std=True)
    mo, so = orc.cgp_predict(fit, Xs, return_std=True)
    assert mean.shape == (m, r) and std.shape == (m,)
    np.testing.assert_allclose(mean, mo, rtol=1e-9, atol=1e-9 * np.abs(mo).max())
    np.testing.assert_allclose(std ** 2, so ** 2, rtol=1e-9, atol=1e-10 * fit["cov_factor"])
    mean2, cov = gp.predict(Xs[:300], return_cov=True)
    mo2, co = orc.cgp_predict(fit, Xs[:300], return_cov=True)
    np.testing.assert_array_equal(mean2, mean[:300])
    np.testing.assert_array_equal(cov, cov.T)                        # mirrored from the lower tiles
    np.testing.assert_allclose(cov, co, rtol=1e-9, atol=1e-10 * fit["cov_factor"])
    np.testing.assert_allclose(np.diag(cov), std[:300] ** 2, rtol=1e-9, atol=1e-11 * fit["cov_factor"])
