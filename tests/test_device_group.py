"""CPU tests of round 5's single-process fan-out (``devices=``): the sharding / merge / gather logic on ``backend='cpu'`` with N
``CpuContext`` members (tests/test_gpu_round5.py runs the same calls on HIP contexts), and the regressions of ADVICE round 4."""
import functools
import threading
import warnings

import numpy as np
import pytest
from sklearn.gaussian_process.kernels import RBF, ConstantKernel as C

import gsum_amd
from gsum_amd import _cpu
from gsum_amd.grid import lml_grid_distributed, shard_range


def _fitted(n=60, r=4):
    X = 0.1 * np.arange(n)[:, None]
    c = np.random.RandomState(0).randn(n, r)
    y = gsum_amd.partials(c, ratio=0.5, ref=1.0, orders=np.arange(r))
    gp = gsum_amd.TruncationGP(kernel=RBF(0.2), ratio=0.5, ref=1.0, center=0, disp=0, df=1, scale=1, optimizer=None, backend="cpu")
    gp.fit(X, y, orders=np.arange(r))
    return gp, X, y


@pytest.mark.parametrize("mode", ["full", "reuse"])
@pytest.mark.parametrize("world", [1, 3, 8])
def test_grid_over_a_group_of_cpu_contexts(mode, world):
    """Every member evaluates exactly its shard (one factorisation per owned theta in mode "reuse", one per owned point in mode
    "full"), the merged surface equals the plain call, with and without the gather step and the prior-scale axis."""
    gp, X, y = _fitted()
    thetas = [np.log([ls]) for ls in np.linspace(0.1, 0.4, 7)]
    ratios = list(np.linspace(0.3, 0.7, 5))
    want = gp.log_marginal_likelihood_grid(thetas, ratios, mode=mode)
    _cpu._cpu_groups.pop(world, None)                       # a fresh group: its members' call counters start at zero
    devices = list(range(world))
    got = gp.log_marginal_likelihood_grid(thetas, ratios, mode=mode, devices=devices)
    assert np.array_equal(got, want)
    grp = _cpu.cpu_group(world)
    per_member = [c.calls["potrf"] for c in grp.contexts]
    if mode == "full":
        expect = [shard_range(35, r, world)[1] - shard_range(35, r, world)[0] for r in range(world)]
    else:
        expect = [shard_range(7, r, world)[1] - shard_range(7, r, world)[0] for r in range(world)]
    assert per_member == expect
    got = gp.log_marginal_likelihood_grid(thetas, ratios, mode=mode, devices=devices, gather="rccl")
    assert np.array_equal(got, want) and grp.get("rccl_gathers") == 1
    want_s = gp.log_marginal_likelihood_grid(thetas[:2], ratios, scales=[0.5, 1.0, 2.0], mode=mode)
    got_s = gp.log_marginal_likelihood_grid(thetas[:2], ratios, scales=[0.5, 1.0, 2.0], mode=mode, devices=devices, gather="rccl")
    assert np.array_equal(got_s, want_s)


def test_group_errors_and_argument_checks():
    gp, X, y = _fitted()
    thetas, ratios = [np.log([0.2])], [0.4, 0.5]
    with pytest.raises(ValueError, match="exclude"):
        gp.log_marginal_likelihood_grid(thetas, ratios, devices=[0, 1], shard=(0, 2))
    with pytest.raises(ValueError, match="gather"):
        gp.log_marginal_likelihood_grid(thetas, ratios, devices=[0, 1], gather="mpi")
    with pytest.raises(ValueError, match="mode"):
        gp.log_marginal_likelihood_grid(thetas, ratios, devices=[0, 1], mode="fast")

    # an exception inside one member's thread reaches the caller
    grp = _cpu.CpuGroup(3)

    def boom(r, ctx):
        if r == 2:
            raise KeyError("member 2")
        return r
    with pytest.raises(KeyError, match="member 2"):
        grp.map(boom)
    assert grp.map(lambda r, ctx: (r, threading.current_thread().name))[0][0] == 0


def test_group_scan_of_the_cpu_backend_matches_its_members():
    """CpuGroup.lml_batch / lml_resident: the C ABI's gsum_lml_batch_multi semantics (block partition, results at their positions)."""
    X = 0.1 * np.arange(50)[:, None]
    Z = np.concatenate([np.random.RandomState(1).randn(50, 3), np.ones((50, 1))], axis=1)
    descs = [gsum_amd.describe_kernel(RBF(0.15 + 0.02 * j), 1) for j in range(7)]
    want = _cpu.CpuContext().lml_batch(descs, X, Z, 1e-10)
    for world in (1, 2, 4, 9):
        grp = _cpu.CpuGroup(world)
        for got in (grp.lml_batch(descs, X, Z, 1e-10), grp.lml_batch(descs, X, Z, 1e-10, gather="rccl")):
            for a, b in zip(got, want):
                assert np.array_equal(a, b)
        grp.set_inputs(X, Z)
        for a, b in zip(grp.lml_resident(descs, 1e-10), want):
            assert np.array_equal(a, b)


def test_predict_over_a_group_of_cpu_contexts():
    gp, X, y = _fitted()
    cg = gp.coeffs_process
    Xs = np.linspace(0, 6, 37)[:, None]
    mean, std = cg.predict(Xs, return_std=True)
    for devices in ([0], [0, 1, 2], "all"):
        m2, s2 = cg.predict(Xs, return_std=True, devices=devices)
        assert np.array_equal(m2, mean) and np.array_equal(s2, std)
    m3, s3 = cg.predict(Xs, return_std=True, devices=[0, 1], Xc=X[::2], y=cg.y_train_[::2])
    m4, s4 = cg.predict(Xs, return_std=True, Xc=X[::2], y=cg.y_train_[::2])
    assert np.array_equal(m3, m4) and np.array_equal(s3, s4, equal_nan=True)
    with pytest.raises(ValueError, match="return_cov"):
        cg.predict(Xs, return_cov=True, devices=[0, 1])
    # more members than new points: empty blocks
    m5 = cg.predict(Xs[:2], devices=list(range(5)))
    assert np.allclose(m5, cg.predict(Xs[:2]), rtol=1e-12, atol=0)          # (host BLAS: a one-point block takes another code path)


# ---- ADVICE round 4 ----------------------------------------------------------------------------------------------------------------
def test_multi_start_fit_leaves_the_warnings_machinery_alone():
    """the restart threads used ``warnings.catch_warnings``, whose save / restore of process-global state is only
    safe with LIFO exits; after a multi-start fit every later warning of the process was swallowed."""
    n = 40
    X = np.linspace(0, 4, n)[:, None]
    y = np.random.RandomState(0).randn(n, 2)
    show_before, filters_before = warnings.showwarning, list(warnings.filters)
    impl_before = getattr(warnings, "_showwarnmsg_impl", None)
    gp = gsum_amd.ConjugateGaussianProcess(kernel=C(1.0, (1e-2, 1e2)) * RBF(0.5, (1e-1, 1e1)), center=0, disp=0, df=1, scale=1,
                                           n_restarts_optimizer=3, random_state=1, backend="cpu")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gp.fit(X, y)
    assert warnings.showwarning is show_before and warnings.filters == filters_before
    assert getattr(warnings, "_showwarnmsg_impl", None) is impl_before
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        warnings.warn("still heard", RuntimeWarning)
    assert len(rec) == 1

    # a start's convergence message is emitted from the calling thread
    seen = []
    gp2 = gsum_amd.ConjugateGaussianProcess(kernel=C(1.0, (1e-2, 1e2)) * RBF(0.5, (1e-1, 1e1)), center=0, disp=0, df=1, scale=1,
                                            n_restarts_optimizer=2, random_state=1, backend="cpu")
    orig = gp2._constrained_optimization

    def noisy(obj, theta0, bounds, warn=None):
        (warn or warnings.warn)("start did not converge", RuntimeWarning)
        seen.append(threading.current_thread() is threading.main_thread())
        return orig(obj, theta0, bounds, warn=warn)
    gp2._constrained_optimization = noisy
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        gp2.fit(X, y)
    assert sum("did not converge" in str(w.message) for w in rec) == 3       # the first start + two restarts
    assert seen.count(False) >= 2                                            # ... of which the restarts ran on worker threads


def test_gather_refuses_a_surface_with_holes():
    """a ``mode="reuse"`` scan fills whole thetas; gathered as the default flat partition it used to come back with
    NaN holes at world > 1.  The rank's own slice is now checked before the gather."""
    gp, X, y = _fitted()
    thetas = [np.log([ls]) for ls in np.linspace(0.1, 0.4, 6)]
    ratios = list(np.linspace(0.3, 0.7, 4))

    def evaluate_as_rank1_of_2(shard=None):          # a rank of a world of 2 without a process group: only the check runs
        return gp.log_marginal_likelihood_grid(thetas, ratios, mode="reuse", shard=(0, 2))
    import gsum_amd.grid as grid
    surface = evaluate_as_rank1_of_2()
    flat_lo, flat_hi = shard_range(surface.size, 0, 2)
    assert np.isnan(surface.reshape(-1)[flat_lo:flat_hi]).any()             # the mismatch ADVICE describes
    with pytest.raises(ValueError, match='partition="theta"'):
        grid._no_holes(surface.reshape(-1)[flat_lo:flat_hi], "flat", 0)
    by_theta = np.moveaxis(surface, 1, 0)
    jlo, jhi = shard_range(len(thetas), 0, 2)
    grid._no_holes(by_theta[jlo:jhi], "theta", 0)                           # the matching partition passes
    # the plain single-process call is unaffected
    fn = functools.partial(gp.log_marginal_likelihood_grid, thetas, ratios, mode="reuse")
    assert np.array_equal(lml_grid_distributed(fn, partition="theta"), gp.log_marginal_likelihood_grid(thetas, ratios, mode="reuse"))


def test_ratio_and_ref_accept_the_override_positionally():
    """the reference's closures are ``lambda X, ratio=ratio: ...`` (models.py:1310, 1315): ``gp.ratio(X, 0.4)`` works."""
    kws = {}
    gp = gsum_amd.TruncationGP(kernel=RBF(0.2), ratio=0.5, ref=2.0, ratio_kws=kws, backend="cpu")
    X = np.zeros((3, 1))
    assert np.array_equal(gp.ratio(X, 0.4), 0.4 * np.ones(3))
    assert np.array_equal(gp.ref(X, 3.0), 3.0 * np.ones(3))
    assert np.array_equal(gp.ratio(X, ratio=0.3), 0.3 * np.ones(3)) and np.array_equal(gp.ratio(X), 0.5 * np.ones(3))
    with pytest.raises(TypeError):
        gp.ratio(X, ref=1.0)
    with pytest.raises(TypeError):
        gp.ratio(X, 0.4, ratio=0.3)
    assert gp.ratio_kws is kws                          # the caller's dict, like models.py:1326


def test_truncation_predict_over_a_group_of_cpu_contexts():
    """``TruncationGP.predict(devices=...)``: new points in blocks, every member conditioning on its own copy of cov(Xc, Xc)."""
    gp, X, y = _fitted()
    Xs = np.linspace(0.05, 5.9, 41)[:, None]
    for kind in ("both", "interp", "trunc"):
        mean, std = gp.predict(Xs, order=2, return_std=True, kind=kind)
        for devices in ([0], [0, 1, 2]):
            m2, s2 = gp.predict(Xs, order=2, return_std=True, kind=kind, devices=devices)
            np.testing.assert_allclose(m2, mean, rtol=1e-11, atol=1e-11)
            np.testing.assert_allclose(s2, std, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(gp.predict(Xs, order=2, kind=kind, devices=[0, 1]), mean, rtol=1e-11, atol=1e-11)
    with pytest.raises(ValueError, match="return_cov"):
        gp.predict(Xs, order=2, return_cov=True, devices=[0, 1])
