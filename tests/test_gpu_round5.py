"""Round-5 GPU tests (``-m gpu``): the device group of the C ABI (gsum_init_multi / gsum_lml_batch_multi / the in-library RCCL gather),
``devices=`` on the model classes, the stream probe of gsum_init, the product library's factor against LAPACK."""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
from sklearn.gaussian_process.kernels import RBF, Matern, WhiteKernel

from conftest import ROOT

pytestmark = pytest.mark.gpu

import gsum_amd  # noqa: E402
from gsum_amd import _lib  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    return gsum_amd.default_context(0)


def _scan_inputs(n, k=5, n_theta=13):
    X = 0.1 * np.arange(n)[:, None]
    rng = np.random.RandomState(5)
    Z = np.concatenate([rng.randn(n, k - 1), np.ones((n, 1))], axis=1)
    descs = [gsum_amd.describe_kernel(RBF(0.15 + 0.01 * j), 1) for j in range(n_theta)]
    return X, Z, descs


@pytest.mark.parametrize("n", [100, 700, 2300])
def test_group_scan_equals_the_one_device_call(ctx, n):
    """gsum_lml_batch_multi / gsum_group_lml_resident over [0] and over every visible GPU: G, sum log diag and info equal the
    one-context call bit for bit, with the host merge and with the in-library RCCL gather (ncclCommInitAll over the group's devices;
    world 1 on a one-GPU box, where the collective still runs through RCCL).  Replaces the serial loop over grid points,
    docs/notebooks/correlated_EFT_publication.ipynb:1457-1459."""
    X, Z, descs = _scan_inputs(n)
    want = ctx.lml_batch(descs, X, Z, 1e-10)
    for devices in ([0], "all"):
        grp = gsum_amd.default_group(devices)
        assert len(grp) == (1 if devices == [0] else gsum_amd.device_count())
        for gather in ("host", "rccl"):
            got = grp.lml_batch(descs, X, Z, 1e-10, gather=gather)
            for a, b in zip(got, want):
                assert np.array_equal(a, b), (devices, gather)
        grp.set_inputs(X, Z)
        got = grp.lml_resident(descs, 1e-10, gather="rccl")
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        assert grp.get("rccl") == 1 and grp.get("rccl_gathers") >= 2
        assert grp.get("devices_used") == min(len(grp), len(descs))


def test_group_threads_on_two_contexts_of_one_device(ctx):
    """The fan-out itself -- one host thread per member, each block written into its positions of the caller's arrays -- on a box
    with one GPU: a group that OWNS two contexts on device 0 (gsum_init_multi with device_ids {0, 0}).  Bit-identical to the
    one-context call; the RCCL gather refuses a device listed twice with a message instead of hanging in ncclCommInitAll."""
    X, Z, descs = _scan_inputs(900, n_theta=9)
    want = ctx.lml_batch(descs, X, Z, 1e-10)
    grp = gsum_amd.HipGroup([0, 0], own=True)
    try:
        assert len(grp) == 2
        got = grp.lml_batch(descs, X, Z, 1e-10)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        assert grp.get("devices_used") == 2
        # fewer descriptors than members: the second block is empty
        got = grp.lml_batch(descs[:1], X, Z, 1e-10)
        assert np.array_equal(got[1], want[1][:1]) and grp.get("devices_used") == 1
        with pytest.raises(RuntimeError, match="twice"):
            grp.lml_batch(descs, X, Z, 1e-10, gather="rccl")
        # a failed evaluation is a value, not an error: a matrix that is not positive definite reports its pivot in every block
        bad = [gsum_amd.describe_kernel(RBF(50.0), 1)] * 4
        info = grp.lml_batch(bad, X, Z, 0.0)[2]
        assert np.all(info > 0) and np.array_equal(info, ctx.lml_batch(bad, X, Z, 0.0)[2])
        # errors name the member
        with pytest.raises(ValueError, match="rank"):
            grp.lml_batch(descs, X[:, [0] * 9], Z, 1e-10)
    finally:
        grp.close()


def test_group_allgather_of_a_row_partitioned_array():
    grp = gsum_amd.default_group("all")
    a = np.random.RandomState(0).randn(37, 3)
    out = grp.allgather(a)
    assert np.array_equal(out, a)
    assert grp.allgather(np.empty((0, 4))).shape == (0, 4)


def _fitted(n, r=4, kernel=None, **kw):
    X = 0.1 * np.arange(n)[:, None]
    K = RBF(0.2)(X) + 1e-10 * np.eye(n)
    c = np.linalg.cholesky(K) @ np.random.RandomState(0).randn(n, r)
    y = gsum_amd.partials(c, ratio=0.5, ref=1.0, orders=np.arange(r))
    gp = gsum_amd.TruncationGP(kernel=kernel or RBF(0.2), ratio=0.5, ref=1.0, center=0, disp=0, df=1, scale=1, optimizer=None, **kw)
    gp.fit(X, y, orders=np.arange(r))
    return gp, X, y


@pytest.mark.parametrize("mode", ["full", "reuse"])
def test_grid_over_devices_equals_the_plain_grid(mode):
    """``log_marginal_likelihood_grid(devices=...)``: the surface of the single-process caller (notebook :1444-1459) from a group
    of GPUs equals the one-device surface bit for bit -- [0], every visible GPU, host merge and RCCL gather, with and without the
    prior-scale axis of BASELINE config 4."""
    gp, X, y = _fitted(500)
    thetas = [np.log([ls]) for ls in np.linspace(0.12, 0.3, 7)]
    ratios = list(np.linspace(0.3, 0.7, 5))
    want = gp.log_marginal_likelihood_grid(thetas, ratios, mode=mode)
    want_s = gp.log_marginal_likelihood_grid(thetas[:2], ratios, scales=[0.5, 1.0, 2.0], mode=mode)
    for devices in ([0], "all"):
        for gather in ("host", "rccl"):
            got = gp.log_marginal_likelihood_grid(thetas, ratios, mode=mode, devices=devices, gather=gather)
            assert np.array_equal(got, want), (devices, gather)
        got = gp.log_marginal_likelihood_grid(thetas[:2], ratios, scales=[0.5, 1.0, 2.0], mode=mode, devices=devices, gather="rccl")
        assert np.array_equal(got, want_s)
    with pytest.raises(ValueError):
        gp.log_marginal_likelihood_grid(thetas, ratios, devices=[0], shard=(0, 1))


def test_predict_over_devices_equals_the_plain_predict():
    """``predict(devices=...)``: new points cut into one block per device, each device holding its own copy of the factor; columns
    of the predictive covariance are independent per new point (models.py:836)."""
    n = 700
    rng = np.random.RandomState(3)
    X = rng.rand(n, 2) * [6.0, 9.0]
    y = rng.randn(n, 3)
    gp = gsum_amd.ConjugateGaussianProcess(kernel=Matern([0.7, 1.3], nu=2.5) + WhiteKernel(1e-6, "fixed"), center=0, disp=0, df=1,
                                           scale=1, optimizer=None).fit(X, y)
    Xs = rng.rand(301, 2) * [6.0, 9.0]
    mean, std = gp.predict(Xs, return_std=True)
    for devices in ([0], "all"):
        m2, s2 = gp.predict(Xs, return_std=True, devices=devices)
        assert np.array_equal(m2, mean) and np.array_equal(s2, std)
        m3 = gp.predict(Xs, devices=devices, Xc=X[::2], y=y[::2])
        assert np.array_equal(m3, gp.predict(Xs, Xc=X[::2], y=y[::2]))
    with pytest.raises(ValueError):
        gp.predict(Xs, return_cov=True, devices=[0])


def test_truncation_predict_over_devices_equals_the_plain_predict():
    """``TruncationGP.predict(devices=...)`` (every kind): blocks of new points, each device conditioning on its own cov(Xc, Xc)."""
    gp, X, y = _fitted(300)
    Xs = np.linspace(0.03, 29.9, 257)[:, None]
    for kind in ("both", "interp", "trunc"):
        mean, std = gp.predict(Xs, order=2, return_std=True, kind=kind)
        for devices in ([0], "all"):
            m2, s2 = gp.predict(Xs, order=2, return_std=True, kind=kind, devices=devices)
            assert np.array_equal(m2, mean) and np.array_equal(s2, std, equal_nan=True)


def test_streams_of_a_context_run_side_by_side(ctx):
    """gsum_init's pairwise stream probe (option "pipes_ok"): the context's four streams overlap pairwise -- also in the second, third,
    ... context of a process, where the runtime's hand-out of hardware queues put two of the four on one queue until gsum_init
    learnt to replace such a stream (profiles/r05_pipe_probe.log).  The lab entry shows the matrix, and what streams created
    later look like (recorded only: that mapping is the runtime's)."""
    assert ctx.get_option("pipes_ok") == 1, ctx.get_option("pipe_overlap_permille")
    assert ctx.get_option("pipe_overlap_permille") >= 500
    lab = gsum_amd.lab_context(0)
    assert lab.get_option("pipes_ok") == 1
    more = [_lib.HipContext(0) for _ in range(3)]
    try:
        assert [c.get_option("pipes_ok") for c in more] == [1, 1, 1]
        heals = [c.get_option("pipe_heals") for c in more]
    finally:
        for c in more:
            c.close()
    m = lab.pipe_probe(extra=2)
    assert m.shape == (6, 6) and np.allclose(m, m.T) and np.all(m[:4, :4] >= 0.5)
    from conftest import record_parity
    record_parity("pipe_probe", min_overlap_of_the_four=float(m[:4, :4].min()), bound=0.5, streams_replaced_in_three_more_contexts=heals,
                  later_streams_min_overlap=[float(m[:4, 4].min()), float(m[:4, 5].min())])


def test_product_library_factor_against_lapack(ctx):
    """gsum_potrf_lower of the PRODUCT library (the building-block tests of test_gpu_parity.py run on the lab build of the same
    sources) against numpy.linalg.cholesky, factor against factor: n = 1000 in full, n = 8192 on sampled rows (LAPACK's factor of
    the 8192 matrix takes seconds on the host).  Replaces models.py:711, 809, 969."""
    for n, rows in ((1000, None), (8192, np.random.RandomState(0).choice(8192, 24, replace=False))):
        X = 0.1 * np.arange(n)[:, None]
        desc = gsum_amd.describe_kernel(RBF(0.2), 1)
        K = RBF(0.2)(X) + 1e-6 * np.eye(n)
        want = np.linalg.cholesky(K)
        A = ctx.kernel_matrix_dev(desc, X, diag_add=1e-6)
        try:
            assert ctx.potrf(A) == 0
            got = A.to_host()
        finally:
            A.free()
        assert np.all(np.triu(got, 1) == 0.0)
        sel = slice(None) if rows is None else np.sort(rows)
        err = np.abs(got[sel] - want[sel]).max()
        assert err < 1e-9, err                       # cond(K) ~ 1e6: two backward-stable factors agree to ~ eps * cond
        back = got[sel] @ got.T - K[sel]
        assert np.abs(back).max() < 1e-12


def test_rccl_c_host_drives_its_devices_concurrently():
    """tests/c_host/shard_host_rccl.c, round 5: one pthread per device runs that device's block (gsum_lml_resident_shard), so the
    devices of a C host work at the same time; the recipe's three in-place ncclAllGather calls follow.  And the same scan through
    the library's own group entry (gsum_lml_batch_multi with GSUM_GATHER_RCCL) from C."""
    from test_host_logic import _build_rccl_host
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "shard_host_rccl")
        res = _build_rccl_host(exe)
        if res is None:
            pytest.skip("gcc or the RCCL headers are not on this box")
        assert res.returncode == 0, res.stderr
        env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "gsum_amd") + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
        run = subprocess.run([exe, "1500", "13"], capture_output=True, text=True, timeout=600, env=env)
        assert run.returncode == 0, run.stdout + run.stderr
        assert "threads" in run.stdout and "group entry: yes" in run.stdout, run.stdout


# ---- kernel trees on the one-workgroup-per-evaluation paths (missing 3) ---------------------------------------------------
TREES = ["RBF(0.2) + RBF(2.5) + WhiteKernel(1e-8)", "C(2.0) * RBF(0.3) + C(0.5) * Matern(0.8, nu=2.5) + WhiteKernel(1e-6)",
         "RBF(0.25) * Matern(1.5, nu=1.5) + WhiteKernel(1e-7)", "RationalQuadratic(0.4, alpha=1.3) + RBF(0.2) + WhiteKernel(1e-6)"]


@pytest.mark.parametrize("expr", TREES)
@pytest.mark.parametrize("n,d", [(5, 1), (97, 2), (128, 1), (300, 1), (700, 2)])
def test_tree_kernels_on_the_fused_paths_equal_the_general_path(ctx, expr, n, d):
    """A Sum / Product tree on k_lml_small<true> (n <= 128) and k_lml_medium<true> (n <= 4096) gives G, sum log diag and info equal,
    bit for bit, to the grouped multi-kernel schedule (k_build_tree + the blocked factorisation) -- the reference accepts any kernel
    (models.py:146-147, 958-960) and its own workloads are 5 ... 20 points on 8000-point grids; mixed calls (flattened and tree
    descriptors in one launch) too."""
    from conftest import tree_kernel
    rng = np.random.RandomState(n)
    X = rng.rand(n, d) * (0.12 * n if d == 1 else np.sqrt(n) * 0.4)
    Z = np.concatenate([rng.randn(n, 3), np.ones((n, 1))], axis=1)
    kern = tree_kernel(expr)
    descs = [gsum_amd.describe_kernel(kern.clone_with_theta(kern.theta + dt), d) for dt in np.linspace(-0.15, 0.15, 5)]
    assert all(dd.is_tree for dd in descs)
    mixed = descs[:2] + [gsum_amd.describe_kernel(RBF(0.3) + WhiteKernel(1e-6), d)] + descs[2:]
    ctx.set_inputs(X, Z)
    ctx.set_option("medium_min_batch", 1)
    try:
        got = ctx.lml_resident(mixed, 1e-10)
        ctx.set_option("small_path", 0)
        ctx.set_option("medium_path", 0)
        want = ctx.lml_resident(mixed, 1e-10)
    finally:
        ctx.set_option("small_path", 1)
        ctx.set_option("medium_path", 1)
        ctx.set_option("medium_min_batch", -1)
    assert np.all(want[2] == 0)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)


def test_tree_kernel_golden_through_the_small_path():
    """tests/golden/tree_kernels.json (values of the reference itself, n = 60): the likelihood of every case through a CALL OF SEVERAL
    evaluations, i.e. k_lml_small<true> (tests/test_gpu_parity.py::test_tree_kernels_golden makes the single calls)."""
    from conftest import load_golden, tree_kernel
    ctx = gsum_amd.default_context(0)
    done = 0
    for case in load_golden("tree_kernels.json"):
        X, y = np.array(case["X"]), np.array(case["y"])
        kern = tree_kernel(case["expr"])
        gp = gsum_amd.ConjugateGaussianProcess(kernel=kern, optimizer=None, nugget=case["nugget"], **case["priors"])
        descs = [gsum_amd.describe_kernel(kern.clone_with_theta(np.array(ev["theta"])), case["dim"]) for ev in case["evals"]]
        assert all(dd.is_tree for dd in descs)
        G, sld, info = ctx.lml_batch(descs * 3, X, gp._rhs(X, y), gp.nugget)
        lml = gp._lml_gram_batch(G, sld, X.shape[0])
        assert np.all(info == 0)
        vtol = max(1e-10, 3e-17 * case["cond"])
        for q, ev in enumerate(case["evals"]):
            assert lml[q] == lml[q + len(descs)] == lml[q + 2 * len(descs)]
            assert lml[q] == pytest.approx(ev["lml"], rel=vtol), case["expr"]
            done += 1
    assert done >= 6


def test_notebook_grid_with_a_tree_kernel_stays_on_the_fast_path():
    """The notebook's 80 x 100 scan (n = 5, 8000 evaluations) with RBF(0.2) + RBF(2.5) + White: within 1.5 x of the flattened
    kernel's time (round 4: a tree fell off the one-workgroup path onto the 256-padded multi-kernel schedule, per evaluation)."""
    import time
    from conftest import load_golden, record_parity
    g = load_golden("notebook_grid.json")
    X, y = np.array(g["X_train"]), np.array(g["y_train"])
    flat = RBF(0.2) + WhiteKernel(g["nugget"], noise_level_bounds="fixed")
    tree = RBF(0.2) + RBF(2.5, length_scale_bounds="fixed") + WhiteKernel(g["nugget"], noise_level_bounds="fixed")
    times = {}
    for name, kern in (("flat", flat), ("tree", tree)):
        gp = gsum_amd.TruncationGP(kernel=kern, ref=g["ref"], ratio=0.5, center=0, disp=0, df=1, scale=1, optimizer=None)
        gp.fit(X, y, orders=np.array(g["orders"]))
        thetas = [[t] for t in np.log(g["ls_vals"])]
        gp.log_marginal_likelihood_grid(thetas, g["ratio_vals"], mode="full")
        best = np.inf
        for _ in range(12):                # (most of a call is host algebra: the minimum of many, host noise only ever adds)
            t0 = time.perf_counter()
            grid = gp.log_marginal_likelihood_grid(thetas, g["ratio_vals"], mode="full")
            best = min(best, time.perf_counter() - t0)
        assert grid.shape == (80, 100) and np.isfinite(grid).all()
        times[name] = best
        if name == "tree":           # spot check against the oracle's scikit-learn arithmetic
            from oracle import gsum_oracle as orc
            for (i, j) in ((0, 0), (36, 39), (79, 99)):
                want = orc.trunc_lml(kern, np.array(thetas[j]), X, y, np.array(g["orders"]), ratio=g["ratio_vals"][i], ref=g["ref"])
                assert abs(grid[i, j] - want) <= 1e-10 * abs(want)
    record_parity("notebook_grid_tree_vs_flat_ms", flat_ms=times["flat"] * 1e3, tree_ms=times["tree"] * 1e3, bound=1.5)
    assert times["tree"] <= 1.5 * times["flat"], times


# ---- O(m) variance, the predictive sweep on k_panel256 ------------------------------------------------------
def test_truncation_std_moves_no_square_matrix(monkeypatch):
    """``TruncationGP.predict(return_std=True)`` and ``underlying_properties(return_std=True)`` used to build the m x m ``K_nn`` on the
    device, copy it over PCIe and take ``np.diag`` (2 GiB at m = 16384 for an O(m) answer; models.py:1443-1482 does the same on the
    host).  Now the diagonal is host arithmetic: every array that crosses the binding during the call is counted here, and the values
    equal ``np.diag`` of the full matrices."""
    gp, X, y = _fitted(400)
    m = 1500
    Xs = np.linspace(0.0, 40.0, m)[:, None]
    ctx = gp.coeffs_process._context()
    moved = []
    for name in ("kernel_matrix", "predict_terms", "forward_gram", "forward_solve", "cho_solve"):
        orig = getattr(ctx, name)

        def counted(*a, _orig=orig, **kw):
            out = _orig(*a, **kw)
            for arr in (out if isinstance(out, tuple) else (out,)):
                if isinstance(arr, np.ndarray):
                    moved.append(arr.nbytes)
            return out
        monkeypatch.setattr(ctx, name, counted)
    for kind in ("both", "interp", "trunc"):
        moved.clear()
        mean, std = gp.predict(Xs, order=2, return_std=True, kind=kind)
        assert sum(moved) < 1 << 20, (kind, sum(moved))                     # < 1 MB (the m x m matrix alone would be 18 MB here)
        mean_c, cov = gp.predict(Xs, order=2, return_cov=True, kind=kind)
        var = np.diag(cov)                      # (the variance is a difference that cancels near the conditioning points: absolute bound)
        np.testing.assert_allclose(std ** 2, var, rtol=1e-9, atol=1e-12 * float(np.abs(var).max()))
        np.testing.assert_array_equal(mean, mean_c)
    moved.clear()
    mu, sd = gp.underlying_properties(Xs, order=1, return_std=True)
    assert sum(moved) == 0
    np.testing.assert_allclose(sd, np.sqrt(np.diag(gp.underlying_properties(Xs, order=1, return_cov=True)[1])), rtol=1e-13)
    cg = gp.coeffs_process
    moved.clear()
    mu, sd = cg.underlying_properties(Xs, return_std=True)
    assert sum(moved) == 0
    np.testing.assert_allclose(sd, np.sqrt(np.diag(cg.cov(Xs))), rtol=1e-13)


@pytest.mark.parametrize("n,m", [(700, 300), (2304, 1100), (2200, 1300)])
def test_predict_sweep_on_panel256_equals_the_three_launch_sweep(n, m):
    """The predictive sweep V^T = K* L^-T with every pair of block columns solved by ONE k_panel256 launch (sibling images rebuilt from
    the factor: k_make_lsib) against the k_panel / K = 128 GEMM / k_panel sweep of rounds 1-4: same arithmetic, same bits --
    column sums of squares, V^T W and the full V^T V (models.py:822-836)."""
    lab = gsum_amd.lab_context(0)
    rng = np.random.RandomState(n)
    X = rng.rand(n, 2) * np.array([0.35, 0.65]) * np.sqrt(n)
    Xs = rng.rand(m, 2) * np.array([0.35, 0.65]) * np.sqrt(n)
    rhs = np.concatenate([rng.randn(n, 5), np.ones((n, 1))], axis=1)
    desc = gsum_amd.describe_kernel(Matern([0.7, 1.3], nu=2.5) + WhiteKernel(1e-6), 2)
    for persist in (0, 1):                     # a factor of the unfused host-enqueued schedule (no sibling images) and of the chain schedule
        lab.set_option("chain_persist", persist)
        L, info = lab.factorize(desc, X, diag_add=1e-10)
        assert info == 0
        try:
            lab.set_option("predict_panel256", 0)
            lab.set_option("predict_split", 0)
            lab.set_option("predict_depth", 1)
            lab.set_option("predict_lookahead", 0)
            want = lab.predict_terms(L, desc, X, Xs, rhs=rhs, want_cov=True)
            lab.set_option("lazy_min_np", 1024)                # (so that the grouping of trailing updates is on at these orders)
            for p256, split, depth in ((1, 0, 1), (0, 1, 2), (1, 1, 2), (1, 1, 3), (1, 1, 4), (1, 0, 4)):
                # ... the two half-sweeps on two streams (rows of the new points never meet), trailing updates grouped 2 / 3 / 4 pairs deep
                lab.set_option("predict_panel256", p256)
                lab.set_option("predict_split", split)
                lab.set_option("predict_depth", depth)
                got = lab.predict_terms(L, desc, X, Xs, rhs=rhs, want_cov=True)
                for a, b in zip(got, want):
                    np.testing.assert_array_equal(a, b)
            # ... and the look-ahead sweep (panels and near updates on the chain stream, one far launch per macro-step beside them)
            lab.set_option("predict_lookahead", 1)
            lab.set_option("predict_panel256", 1)
            lab.set_option("predict_split", 0)
            lab.set_option("predict_depth", 2)
            for rows, depth in ((m, 2), (max(1024, m - 37), 2), (m, 3), (m, 4)):
                lab.set_option("predict_depth", depth)
                got = lab.predict_terms(L, desc, X, Xs[:rows], rhs=rhs, want_cov=True)
                lab.set_option("predict_lookahead", 0)
                lab.set_option("predict_depth", 1)
                base = lab.predict_terms(L, desc, X, Xs[:rows], rhs=rhs, want_cov=True) if rows >= 1024 and rows != m else want
                lab.set_option("predict_lookahead", 1)
                if rows == m or rows >= 1024:
                    for a, b in zip(got, base):
                        np.testing.assert_array_equal(a, b)
        finally:
            lab.set_option("predict_panel256", 1)
            lab.set_option("predict_split", 0)
            lab.set_option("predict_depth", 2)
            lab.set_option("predict_lookahead", 1)
            lab.set_option("lazy_min_np", 4352)
            lab.set_option("chain_persist", -1)
            L.free()


# ---- several right-hand-side sets in one call -------------------------------------------------------------
@pytest.mark.parametrize("n,n_thetas", [(60, 9), (300, 40), (1100, 7), (2300, 30)])
def test_resident_sets_equal_one_call_per_set(ctx, n, n_thetas):
    """gsum_set_inputs_sets + gsum_lml_resident_sets: evaluation i reads right-hand-side set set_of[i] -- on every path (one
    workgroup per evaluation for n <= 128 and for many evaluations of a medium order, the grouped schedule, one or two evaluations
    alone) the results equal, bit for bit, one gsum_set_inputs + gsum_lml_resident call per set."""
    rng = np.random.RandomState(n)
    X = 0.1 * np.arange(n)[:, None]
    n_sets, k = 4, 5
    Zs = np.concatenate([rng.randn(n_sets, n, k - 1), np.ones((n_sets, n, 1))], axis=2)
    descs = [gsum_amd.describe_kernel(RBF(0.15 + 0.1 * j / n_thetas), 1) for j in range(n_thetas)]
    set_of = rng.randint(0, n_sets, size=n_thetas)
    want = [np.empty((n_thetas, k, k)), np.empty(n_thetas), np.empty(n_thetas, dtype=np.int64)]
    for s_ in range(n_sets):
        pick = np.flatnonzero(set_of == s_)
        if not len(pick):
            continue
        ctx.set_inputs(X, Zs[s_])
        got = ctx.lml_resident([descs[j] for j in pick], 1e-10)
        for w, g in zip(want, got):
            w[pick] = g
    ctx.set_inputs_sets(X, Zs)
    got = ctx.lml_resident_sets(descs, set_of, 1e-10)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
    # the device group: every device holds all sets, the descriptors are partitioned (gsum_group_lml_resident_sets)
    for devices in ([0], "all"):
        grp = gsum_amd.default_group(devices)
        grp.set_inputs_sets(X, Zs)
        for gather in ("host", "rccl"):
            for a, b in zip(grp.lml_resident_sets(descs, set_of, 1e-10, gather=gather), want):
                np.testing.assert_array_equal(a, b)
    # one or two evaluations: the single-factorisation schedule reads its set too
    got2 = ctx.lml_resident_sets(descs[:2], set_of[:2], 1e-10)
    for a, b in zip(got2, want):
        np.testing.assert_array_equal(a, b[:2])
    with pytest.raises(ValueError, match="set"):
        ctx.lml_resident_sets(descs[:3], [0, n_sets, 1], 1e-10)
    # the plain call after a sets call reads set 0
    got3 = ctx.lml_resident(descs[:5], 1e-10)
    ctx.set_inputs(X, Zs[0])
    for a, b in zip(got3, ctx.lml_resident(descs[:5], 1e-10)):
        np.testing.assert_array_equal(a, b)


def test_two_cohorts_per_group_equal_single_evaluations():
    """Calls of many rounds run two cohorts of evaluations per group, half a round apart on the group's one chain stream (option
    wave_cohorts; the last small far updates on that stream too): every evaluation equals, bit for bit, the same evaluation of a call
    with one cohort and of a call of its own -- several orders, ragged call sizes, right-hand-side sets, a failing evaluation inside."""
    lab = gsum_amd.lab_context(0)
    rng = np.random.RandomState(11)
    try:
        for n, N, size in ((1100, 61, 3), (2304, 50, 2), (4096, 40, 2)):
            X = 0.1 * np.arange(n)[:, None]
            Zs = np.concatenate([rng.randn(3, n, 4), np.ones((3, n, 1))], axis=2)
            descs = [gsum_amd.describe_kernel(RBF(0.15 + 0.1 * j / N), 1) for j in range(N)]
            descs[N // 2] = gsum_amd.describe_kernel(RBF(40.0), 1)            # not positive definite with a zero nugget ... (info > 0 below)
            set_of = rng.randint(0, 3, size=N)
            lab.set_option("medium_path", 0)
            lab.set_option("wave_size", size)
            lab.set_option("wave_cohort_min", 2)
            lab.set_inputs_sets(X, Zs)
            lab.set_option("wave_cohorts", 1)
            want = lab.lml_resident_sets(descs, set_of, 0.0 if n == 1100 else 1e-10)
            lab.set_option("wave_cohorts", 2)
            got = lab.lml_resident_sets(descs, set_of, 0.0 if n == 1100 else 1e-10)
            assert lab.get_option("wave_streams") == 4
            ok = want[2] == 0                                                  # (G and sld are undefined where info > 0)
            np.testing.assert_array_equal(got[2], want[2])
            np.testing.assert_array_equal(got[0][ok], want[0][ok])
            np.testing.assert_array_equal(got[1][ok], want[1][ok])
            if n == 1100:
                assert want[2][N // 2] > 0 and np.count_nonzero(want[2]) >= 1
            one = lab.lml_resident_sets(descs[5:6], set_of[5:6], 0.0 if n == 1100 else 1e-10)
            for a, b in zip(one, want):
                np.testing.assert_array_equal(a[0], b[5])
    finally:
        lab.set_option("medium_path", 1)
        lab.set_option("wave_size", 8)
        lab.set_option("wave_cohort_min", 4)
        lab.set_option("wave_cohorts", 2)
        lab.set_option("release_scratch", 1)


@pytest.mark.parametrize("n,m,k", [(300, 77, 3), (1500, 640, 16)])
def test_predict_var_is_predict_terms_on_the_held_right_hand_sides(ctx, n, m, k):
    """gsum_predict_var (the name and signature SURVEY.md 8b gives the predictive pieces of models.py:822-836): after forward_gram on
    a factor, the column sums of squares and V^T W for the right-hand sides the factor holds -- bit for bit gsum_predict_terms' with the
    same right-hand sides handed over again; without a solved right-hand side V^T W is refused, the sums of squares are not."""
    rng = np.random.RandomState(n + k)
    X = rng.rand(n, 2) * np.array([0.35, 0.65]) * np.sqrt(n)
    Xs = rng.rand(m, 2) * np.array([0.35, 0.65]) * np.sqrt(n)
    rhs = rng.randn(n, k)
    desc = gsum_amd.describe_kernel(Matern([0.7, 1.3], nu=2.5) + WhiteKernel(1e-6), 2)
    L, info = ctx.factorize(desc, X, diag_add=1e-10)
    assert info == 0
    try:
        css0, none = ctx.predict_var(L, desc, X, Xs)
        assert none is None
        with pytest.raises(ValueError, match="solved"):
            ctx.predict_var(L, desc, X, Xs, want_vtw=True)
        ctx.forward_gram(L, rhs)
        css, vtw = ctx.predict_var(L, desc, X, Xs, want_vtw=True)
        want_css, want_vtw, _ = ctx.predict_terms(L, desc, X, Xs, rhs=rhs)
        assert np.array_equal(css, want_css) and np.array_equal(css0, want_css)
        assert vtw.shape == (m, _lib.GSUM_MAX_RHS)
        assert np.array_equal(vtw[:, :k], want_vtw) and not vtw[:, k:].any()
        # against scipy on the host copy of the factor (the oracle's own formulation, models.py:822-836)
        from scipy.linalg import solve_triangular
        Lh = np.tril(L.to_host())
        V = solve_triangular(Lh, ctx.kernel_matrix(desc, X, Xs), lower=True)
        np.testing.assert_allclose(css, np.einsum("ij,ij->j", V, V), rtol=1e-10)
        np.testing.assert_allclose(vtw[:, :k], V.T @ solve_triangular(Lh, rhs, lower=True), rtol=1e-9, atol=1e-10)
    finally:
        L.free()


@pytest.mark.parametrize("n,kind", [(700, "flat"), (1500, "tree"), (2300, "flat"), (4400, "flat")])
def test_single_gradient_evaluation_forms_are_bit_identical(n, kind):
    """One value + gradient evaluation alone (gsum_lml_grad; models.py:957-958, 1041-1056), round-5 form against round-4 form: the U = L^-T sweep's
    launches enqueued between the factorisation's outer steps (gs_potrf_chain's step hook) instead of behind all of them, the kernel-gradient
    contractions split (Q_p beside the R^-1 product, the traces from stored triangles of dR_p) instead of fused, the sweep's trailing updates
    paired.  Same kernels, same terms in the same order: G, sum log diag, traces and H equal bit for bit -- also against the batch path, which
    keeps the fused contraction."""
    from sklearn.gaussian_process.kernels import ConstantKernel as C
    lab = gsum_amd.lab_context(0)
    rng = np.random.RandomState(n)
    d = 2 if kind == "tree" else 1
    X = 0.1 * np.arange(n)[:, None] if d == 1 else rng.rand(n, 2) * np.array([0.35, 0.65]) * np.sqrt(n)
    Z = np.concatenate([rng.randn(n, 5), np.ones((n, 1))], axis=1)
    kern = (C(1.3) * RBF(0.2) + WhiteKernel(1e-8)) if kind == "flat" else (C(0.8) * RBF([0.6, 1.1]) + C(0.3) * Matern(1.4, nu=1.5) + WhiteKernel(1e-7))
    desc, prm = gsum_amd.describe_kernel(kern, d), gsum_amd.kernels.describe_gradient(kern, d)
    names = ("grad_interleave", "grad_split", "grad_lazy_chain")
    aborts = lab.get_option("chain_aborts")            # (earlier tests provoke give-ups on this context on purpose)
    try:
        for name in names:
            lab.set_option(name, 0)
        want = lab.lml_grad(desc, prm, X, Z, 1e-10)
        assert want[2] == 0
        for combo in ((1, 0, 0), (0, 1, 0), (1, 1, 1)):
            for name, v in zip(names, combo):
                lab.set_option(name, v)
            got = lab.lml_grad(desc, prm, X, Z, 1e-10)
            for a, b in zip(got, want):
                assert np.array_equal(np.asarray(a), np.asarray(b)), combo
        batch = lab.lml_grad_batch([desc, desc], [prm, prm], X, Z, 1e-10)
        for a, b in zip(batch, want):
            assert np.array_equal(np.asarray(a)[1], np.asarray(b))
        assert lab.get_option("chain_aborts") == aborts
    finally:
        for name in names:
            lab.set_option(name, 1)


@pytest.mark.parametrize("n,d,kind", [(5, 1, "flat"), (20, 1, "flat"), (33, 2, "tree"), (64, 2, "flat"), (64, 1, "tree"), (100, 2, "tree"), (128, 1, "flat")])
def test_small_gradient_kernel_against_the_general_path(n, d, kind):
    """k_grad_small (n <= 128: value + gradient pieces of a call in ONE launch, one workgroup per kernel -- the reference's own sizes, where
    fit() drives L-BFGS with log_marginal_likelihood(theta, eval_gradient=True): models.py:634-640, 957-958, 1041-1056) against the general
    path (option small_path = 0): G, sum log diag and info bit for bit (the value path's code), traces and H_p to rounding; a call of several
    kernels equals the single calls bit for bit; a matrix that is not positive definite reports its info and zero pieces."""
    from sklearn.gaussian_process.kernels import ConstantKernel as C, RationalQuadratic
    lab = gsum_amd.lab_context(0)
    rng = np.random.RandomState(100 * n + d)
    X = rng.rand(n, d) * (2.0 + 0.05 * n)
    Z = np.concatenate([rng.randn(n, 4), np.ones((n, 1))], axis=1)
    if kind == "flat":
        kern = C(1.4) * Matern(0.7 if d == 1 else [0.6, 1.1], nu=2.5) + WhiteKernel(1e-4) + C(0.2)
    else:
        kern = C(0.9) * RBF(0.8 if d == 1 else [0.7, 1.2]) + C(0.4) * RationalQuadratic(length_scale=1.3, alpha=0.8) + WhiteKernel(1e-4)
    kernels = [kern.clone_with_theta(kern.theta + 0.05 * i) for i in range(5)]
    descs = [gsum_amd.describe_kernel(k, d) for k in kernels]
    prms = [gsum_amd.kernels.describe_gradient(k, d) for k in kernels]
    try:
        lab.set_option("small_path", 0)
        want = [lab.lml_grad(dd, pp, X, Z, 1e-10) for dd, pp in zip(descs, prms)]
        lab.set_option("small_path", 1)
        got = [lab.lml_grad(dd, pp, X, Z, 1e-10) for dd, pp in zip(descs, prms)]
        batch = lab.lml_grad_batch(descs, prms, X, Z, 1e-10)
    finally:
        lab.set_option("small_path", 1)
    for i, (g, wv) in enumerate(zip(got, want)):
        assert g[2] == wv[2] == 0
        assert np.array_equal(g[0], wv[0]) and g[1] == wv[1]                     # G and sum log diag: the value path's own code
        cond = np.linalg.cond(kernels[i](X) + 1e-10 * np.eye(n))
        tol = 1e-14 * cond + 1e-12
        np.testing.assert_allclose(g[3], wv[3], rtol=tol, atol=tol * np.abs(wv[3]).max())
        np.testing.assert_allclose(g[4], wv[4], rtol=tol, atol=tol * np.abs(wv[4]).max())
        for a, b in zip(batch, g):
            assert np.array_equal(np.asarray(a)[i], np.asarray(b))
    # not positive definite: a duplicated point without any jitter
    Xd = X.copy()
    Xd[-1] = Xd[0]
    bad = gsum_amd.describe_kernel(C(1.0) * RBF(0.8 if d == 1 else [0.7, 1.2]), d)
    bp = gsum_amd.kernels.describe_gradient(C(1.0) * RBF(0.8 if d == 1 else [0.7, 1.2]), d)
    G, sld, info, tr, H = lab.lml_grad(bad, bp, Xd, Z, 0.0)
    assert info > 0 and not tr.any() and not H.any()


def test_equal_small_inputs_are_not_uploaded_again(ctx):
    """Objective evaluations of fit() hand the same points and right-hand sides over every time (models.py:634-640): inputs of up to 256 KB
    are remembered on the host and an equal upload is skipped (read-back "uploads_skipped"); a changed byte, another shape or a large input is
    copied, and results never depend on it."""
    rng = np.random.RandomState(11)
    n = 40
    X = rng.rand(n, 2) * 4
    Z = np.concatenate([rng.randn(n, 3), np.ones((n, 1))], axis=1)
    desc = gsum_amd.describe_kernel(Matern([0.7, 1.3], nu=2.5) + WhiteKernel(1e-4), 2)
    first = ctx.lml_batch([desc], X, Z, 1e-10)
    s0 = ctx.get_option("uploads_skipped")
    again = ctx.lml_batch([desc], X, Z, 1e-10)
    assert ctx.get_option("uploads_skipped") == s0 + 2                    # X and Z
    for a, b in zip(first, again):
        assert np.array_equal(a, b)
    Z2 = Z.copy()
    Z2[7, 1] += 1.0
    changed = ctx.lml_batch([desc], X, Z2, 1e-10)
    assert ctx.get_option("uploads_skipped") == s0 + 3 and not np.array_equal(changed[0], first[0])      # X skipped, Z copied
    back = ctx.lml_batch([desc], X, Z, 1e-10)
    assert np.array_equal(back[0], first[0]) and ctx.get_option("uploads_skipped") == s0 + 4
    # another entry point writes the same operator-level set: the memory follows it
    K = ctx.kernel_matrix(desc, X[:10])
    assert K.shape == (10, 10)
    after = ctx.lml_batch([desc], X, Z, 1e-10)
    assert np.array_equal(after[0], first[0])
    big = rng.rand(20000, 2)                                              # 320 KB: not remembered
    s1 = ctx.get_option("uploads_skipped")
    ctx.kernel_matrix(desc, big[:3], big)
    ctx.kernel_matrix(desc, big[:3], big)
    assert ctx.get_option("uploads_skipped") <= s1 + 2


def test_order_32768_property_and_the_refused_orders(ctx):
    """Twice BASELINE's largest configuration (an 8.6-GB augmented matrix, 128 outer steps on the deep-grouped single-factorisation schedule):
    with right-hand sides taken from K itself, G = Z^T K^-1 Z must return K[cols][:, cols] (tools/gpu_large_order.py shows the same at 40960).
    Orders whose padded matrix would have 2^31 elements or more were never exercised and are refused with a message, not attempted."""
    rng = np.random.RandomState(8)
    n = 32768
    X = rng.rand(n, 2) * 58.0
    kern = Matern(length_scale=1.0, nu=2.5) + WhiteKernel(1e-2, noise_level_bounds="fixed")
    cols = np.array([0, 1, 127, 128, 16383, 32767])
    Z = kern(X, X[cols])
    Z[cols, np.arange(len(cols))] += 1e-2
    desc = gsum_amd.describe_kernel(kern, 2)
    G, sld, info = ctx.lml_batch([desc], X, Z, 0.0)
    assert info[0] == 0 and np.isfinite(sld[0])
    np.testing.assert_allclose(G[0], Z[cols], rtol=0, atol=1e-11)
    ctx.set_option("release_scratch", 1)
    big = np.zeros((47000, 1))
    with pytest.raises(ValueError, match="out of range"):
        ctx.lml_batch([gsum_amd.describe_kernel(RBF(1.0), 1)], big, np.ones((47000, 1)), 1e-10)
    with pytest.raises((ValueError, RuntimeError), match="out of range"):
        ctx.kernel_matrix_dev(gsum_amd.describe_kernel(RBF(1.0), 1), big)


def test_partial_diagonal_block_every_count_of_micro_blocks():
    """The one-block kernels factorise only the micro-blocks that hold points (gs_d2_wave<.., PARTIAL>: the reference's own orders are 5-20 of a
    128 x 128 block, the rest is identity padding) and write what the skipped steps would have left behind.  Every count of valid micro-blocks,
    orders on both sides of every boundary: G, sum log diag and info of the one-block path equal the general path's bit for bit (a call of one
    and a call of several evaluations), the gradient pieces agree to rounding, a pivot that fails in the LAST factorised micro-block is found."""
    from sklearn.gaussian_process.kernels import ConstantKernel as C
    lab = gsum_amd.lab_context(0)
    kern = C(1.3) * Matern(0.6, nu=2.5) + WhiteKernel(1e-6)
    desc, prm = gsum_amd.describe_kernel(kern, 1), gsum_amd.kernels.describe_gradient(kern, 1)
    try:
        for n in (1, 2, 5, 15, 16, 17, 31, 32, 33, 47, 48, 49, 64, 65, 79, 80, 81, 96, 97, 111, 112, 113, 127, 128):
            rng = np.random.RandomState(n)
            X = np.sort(rng.rand(n))[:, None] * (0.5 * n + 1.0)
            Z = np.concatenate([rng.randn(n, 3), np.ones((n, 1))], axis=1)
            lab.set_option("small_path", 0)
            want = lab.lml_batch([desc], X, Z, 1e-10)
            gwant = lab.lml_grad(desc, prm, X, Z, 1e-10)
            lab.set_option("small_path", 1)
            got = lab.lml_batch([desc], X, Z, 1e-10)
            got3 = lab.lml_batch([desc, desc, desc], X, Z, 1e-10)
            ggot = lab.lml_grad(desc, prm, X, Z, 1e-10)
            for a, b, c in zip(got, want, got3):
                assert np.array_equal(a, b) and np.array_equal(np.asarray(c)[2], np.asarray(b)[0]), n
            assert np.array_equal(ggot[0], gwant[0]) and ggot[1] == gwant[1] and ggot[2] == gwant[2] == 0, n
            if n > 1:
                np.testing.assert_allclose(ggot[3], gwant[3], rtol=1e-9, atol=1e-9 * np.abs(gwant[3]).max(), err_msg=str(n))
                np.testing.assert_allclose(ggot[4], gwant[4], rtol=1e-9, atol=1e-9 * np.abs(gwant[4]).max(), err_msg=str(n))
        # a duplicated LAST point without jitter: the failing pivot sits in the last factorised micro-block (n = 20: column 19, micro-block 1 of 2)
        for n in (20, 33):
            X = np.arange(n, dtype=float)[:, None] * 0.7
            X[-1] = X[-2]
            bad = gsum_amd.describe_kernel(C(1.0) * RBF(1.0), 1)
            info_s = lab.lml_batch([bad], X, np.ones((n, 1)), 0.0)[2]
            lab.set_option("small_path", 0)
            info_g = lab.lml_batch([bad], X, np.ones((n, 1)), 0.0)[2]
            lab.set_option("small_path", 1)
            assert info_s[0] == info_g[0] == n, (n, info_s, info_g)
    finally:
        lab.set_option("small_path", 1)


@pytest.mark.parametrize("n", [5, 300, 1100])
def test_an_empty_batch_is_no_work(ctx, n):
    """No evaluations (an empty list of thetas, a rank with no share of a grid): nothing is launched, empty outputs, and the context carries on
    -- on the one-block path, the one-workgroup-per-evaluation path and the grouped schedule alike."""
    rng = np.random.RandomState(n)
    X = np.sort(rng.rand(n))[:, None] * n * 0.3
    Z = np.concatenate([rng.randn(n, 2), np.ones((n, 1))], axis=1)
    desc = gsum_amd.describe_kernel(RBF(0.5) + WhiteKernel(1e-4), 1)
    G, sld, info = ctx.lml_batch([], X, Z, 1e-10)
    assert G.shape == (0, 3, 3) and sld.shape == (0,) and info.shape == (0,)
    ctx.set_inputs(X, Z)
    G, sld, info = ctx.lml_resident([], 1e-10)
    assert G.shape == (0, 3, 3) and sld.size == 0
    one = ctx.lml_resident([desc], 1e-10)
    assert one[2][0] == 0 and np.array_equal(one[0], ctx.lml_batch([desc], X, Z, 1e-10)[0])
    gp = gsum_amd.TruncationGP(kernel=RBF(0.5) + WhiteKernel(1e-4), ratio=0.5, ref=1.0, optimizer=None, center=0, disp=0, df=1, scale=1)
    y = gsum_amd.partials(Z[:, :2], ratio=0.5, ref=1.0, orders=np.arange(2))
    gp.fit(X, y, orders=np.arange(2))
    assert gp.log_marginal_likelihood_grid([], [0.4, 0.6], mode="full").shape == (2, 0)
    assert gp.log_marginal_likelihood_grid([gp.coeffs_process.kernel_.theta], [], mode="reuse").shape == (0, 1)
