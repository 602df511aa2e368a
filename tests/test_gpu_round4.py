"""Round-4 GPU tests (``-m gpu``): the RCCL C host, BASELINE config 5 over all eight shards, achieved relative errors of the grids."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu

import gsum_amd  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    return gsum_amd.default_context(0)


@pytest.fixture(scope="module")
def lab():
    """A context on the LAB build of the library (include/gsum_hip_debug.h: schedule switches, test hooks)."""
    return gsum_amd.lab_context(0)


@pytest.mark.parametrize("n,n_theta", [(600, 11), (2048, 8)])
def test_rccl_c_host_gathers_the_sharded_scan(n, n_theta):
    """INTEGRATION.md's multi-GPU recipe from a C host that owns a REAL RCCL communicator (tests/c_host/shard_host_rccl.c):
    ncclCommInitAll over the visible devices, every rank's gsum_lml_resident_shard slice staged to its GPU, the three in-place
    ncclAllGather calls of the recipe, every rank's gathered arrays bit-identical to the unsharded gsum_lml_resident call.  World =
    the GPUs the box shows (1 here, 8 on a node: same binary)."""
    from test_host_logic import _build_rccl_host
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "shard_host_rccl")
        res = _build_rccl_host(exe)
        if res is None:
            pytest.skip("gcc or the RCCL headers are not installed")
        assert res.returncode == 0, res.stderr
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        run = subprocess.run([exe, str(n), str(n_theta)], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "gathered == unsharded on every rank: yes" in run.stdout, run.stdout
    assert "(RCCL, ncclCommInitAll" in run.stdout and "group entry: yes" in run.stdout


@pytest.mark.parametrize("variant", ["b20", "b3"])
def test_single_factorisation_keeps_its_chain_after_a_batch_call(variant):
    """Found by bench.py's single-evaluation leg in round 4: in a process whose FIRST call was a batch, the groups' high-priority chain
    streams were created before slot 0's two, the runtime's four hardware queues of that priority were oversubscribed, the chain
    kernel's stream came to share a queue with its partner's, the two-stream probe (main vs auxiliary) passed -- and the first single
    factorisation ran into the 1-s time-out and fell back to the host-enqueued schedule for the rest of the process (6.7 ms instead of
    5.3 at n = 8192, plus the lost second).  Now the first two groups borrow slot 0's streams and the probe is the schedule's own
    triangle.  A process of its own: the order of stream creation is the scenario."""
    import json
    import sys
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_chain_abort_repro.py"), variant], capture_output=True,
                         text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    rows = [json.loads(line) for line in run.stdout.splitlines() if line.startswith("{")]
    assert len(rows) == 2, run.stdout
    for row in rows:
        assert row["aborts"] == 0 and row["probe"] == 1 and row["persist"] == -1, row
        assert row["wall_ms"] < 500.0, row


def test_chain_step_table_follows_option_changes_on_the_same_workspace(lab):
    """the persistent chain's per-step table fbwant[] used to be uploaded AFTER k_chain was launched, by a copy that
    nothing ordered against the launch; it is re-made whenever the window size or the far-update pairing changes.  It now goes out in
    stream order ahead of the launch, from a buffer the matrix object owns.  Alternating (chain_rows, chain_lazy) on ONE workspace
    matrix -- every switch re-makes the table -- must keep every result bit-identical to the host-enqueued schedule's, with no
    time-out (a stale table would end in a wrong factor or a 1-s give-up)."""
    ctx = lab
    from sklearn.gaussian_process.kernels import RBF
    n = 4096
    rng = np.random.RandomState(4)
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([rng.randn(n, 5), np.ones((n, 1))], axis=1)
    desc = gsum_amd.describe_kernel(RBF(0.21), 1)
    ctx.set_inputs(X, Z)
    try:
        ctx.set_option("chain_persist", 0)
        ref = ctx.lml_resident([desc], 1e-10)
        assert ref[2][0] == 0
        ctx.set_option("chain_persist", 1)
        aborts = ctx.get_option("chain_aborts")
        for rows, lazy in ((512, 0), (256, 2), (512, 1), (256, 0), (512, 2), (512, 0), (256, 1), (512, -1)) * 2:
            ctx.set_option("chain_rows", rows)
            ctx.set_option("chain_lazy", lazy)
            got = ctx.lml_resident([desc], 1e-10)
            for a, b in zip(got, ref):
                np.testing.assert_array_equal(a, b)
        assert ctx.get_option("chain_aborts") == aborts and ctx.get_option("chain_persist") == 1
    finally:
        ctx.set_option("chain_persist", -1)
        ctx.set_option("chain_rows", 512)
        ctx.set_option("chain_lazy", -1)


def test_factorize_recovers_from_a_chain_give_up(lab):
    """a give-up of the single-factorisation schedule inside gsum_potrf_lower destroys the matrix; the binding's
    ``factorize`` (what fit / predict / the reuse grid call) rebuilds it and factorises once more on the host-enqueued schedule, so
    a fit does not fail the first time a time-out happens.  The give-up is forced with the test hook."""
    ctx = lab
    from sklearn.gaussian_process.kernels import RBF
    from gsum_amd._lib import ChainAborted
    n = 2048
    X = 0.1 * np.arange(n)[:, None]
    desc = gsum_amd.describe_kernel(RBF(0.2), 1)
    try:
        ctx.set_option("chain_persist", 1)
        K0, info0 = ctx.factorize(desc, X, diag_add=1e-10)
        L0 = K0.to_host()
        K0.free()
        aborts = ctx.get_option("chain_aborts")
        ctx.set_option("chain_test_abort", 3)
        K = ctx.kernel_matrix_dev(desc, X, diag_add=1e-10)
        with pytest.raises(ChainAborted):
            ctx.potrf(K)
        K.free()
        ctx.set_option("chain_persist", 1)
        ctx.set_option("chain_test_abort", 2)
        K1, info1 = ctx.factorize(desc, X, diag_add=1e-10)                  # gives up once inside, rebuilds, succeeds
        assert info0 == info1 == 0
        np.testing.assert_array_equal(K1.to_host(), L0)
        K1.free()
        assert ctx.get_option("chain_aborts") == aborts + 2
    finally:
        ctx.set_option("chain_persist", -1)


def test_config5_all_eight_shards_against_the_reference():
    """BASELINE configs[4] (n = 16384 2-D points, Matern-5/2 + White, 8 curves, predictive variance "on 8 GPUs"): the configuration's
    16384 new points are 8 shards of 2048, one per GPU (gsum_amd.grid.predict_distributed: shard_range over the new points).  Round 3
    pinned 16 probes, all inside rank 0's shard; tests/golden/s5_predict.json now holds the reference's fit -> predict(return_std)
    (models.py:671-738, 753-845) at 32 probes in EVERY shard (256 in all).  Each shard is predicted as its rank would predict it --
    predict(Xs[lo:hi]) on the whole 2048-point block -- and compared at its probes: mean 1e-9 of the largest mean, variance
    1e-10 * cov_factor, with and without pred_noise."""
    from conftest import record_parity
    from sklearn.gaussian_process.kernels import Matern, WhiteKernel
    d = load_golden("s5_predict.json")
    sp = d["shard_probes"]
    n, r = d["n"], d["r"]
    side = np.array(d["side"])
    X = np.random.RandomState(0).rand(n, 2) * side
    y = np.random.RandomState(2).randn(n, r)
    Xs_all = np.random.RandomState(1).rand(sp["m_all"], 2) * side
    kern = Matern(length_scale=d["length_scale"], nu=2.5) + WhiteKernel(d["white"], noise_level_bounds="fixed")
    gp = gsum_amd.ConjugateGaussianProcess(kernel=kern, center=0, disp=0, df=1, scale=1, optimizer=None)
    gp.fit(X, y)
    assert gp.cov_factor_ == pytest.approx(d["cov_factor"], rel=1e-10)
    idx = np.array(sp["index"])
    want_mean, want_var, want_var_n = np.array(sp["mean"]), np.array(sp["std"]) ** 2, np.array(sp["std_pred_noise"]) ** 2
    scale = np.abs(want_mean).max()
    worst = dict(mean=0.0, var=0.0, var_noise=0.0)
    per_shard = {}
    for rank in range(sp["shards"]):
        lo, hi = gsum_amd.shard_range(sp["m_all"], rank, sp["shards"])
        mine = (idx >= lo) & (idx < hi)
        assert mine.sum() == sp["per_shard"]
        mean, std = gp.predict(Xs_all[lo:hi], return_std=True)                      # the rank's whole block, as predict_distributed calls it
        _, std_n = gp.predict(Xs_all[lo:hi], return_std=True, pred_noise=True)
        loc = idx[mine] - lo
        em = float(np.max(np.abs(mean[loc] - want_mean[mine])) / scale)
        ev = float(np.max(np.abs(std[loc] ** 2 - want_var[mine])) / d["cov_factor"])
        evn = float(np.max(np.abs(std_n[loc] ** 2 - want_var_n[mine])) / d["cov_factor"])
        per_shard[f"shard{rank}"] = dict(mean=em, var=ev, var_noise=evn)
        assert em <= 1e-9 and ev <= 1e-10 and evn <= 1e-10, (rank, em, ev, evn)
        worst = dict(mean=max(worst["mean"], em), var=max(worst["var"], ev), var_noise=max(worst["var_noise"], evn))
    record_parity("config5_s5_eight_shards_vs_reference", probes=int(len(idx)), max_mean_err_over_max_mean=worst["mean"],
                  max_var_err_over_cov_factor=worst["var"], max_var_noise_err_over_cov_factor=worst["var_noise"], per_shard=per_shard)


def test_truncation_cov_is_built_and_scaled_on_the_device(ctx):
    """TruncationProcess.cov (models.py:1343-1348) = ref_i ref_j S(ratio_i ratio_j) cov_factor kernel_ij with position-dependent
    ratio and ref and excluded orders: one device build + one device scaling (gsum_kernel_build_series), against the reference's
    array expression evaluated with numpy on sampled rows -- one- and two-argument forms, fitted and unfitted process."""
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C
    rng = np.random.RandomState(8)
    n, m, r = 4096, 700, 5
    X = np.sort(rng.rand(n))[:, None] * 40.0
    Xp = rng.rand(m, 1) * 40.0
    kern = C(1.7) * RBF(0.9) + WhiteKernel(1e-5, noise_level_bounds="fixed")
    ratio = lambda Z, q=0.4: q + 0.2 * np.sin(Z[:, 0] / 7.0) ** 2          # noqa: E731
    ref = lambda Z: 3.0 + 0.1 * Z[:, 0]                                       # noqa: E731
    gp = gsum_amd.TruncationGP(kernel=kern, ratio=ratio, ref=ref, excluded=[2], center=0, disp=0, df=3, scale=1.3, optimizer=None)
    rows = np.array([0, 1, 17, 2048, 4095])

    def want(Xa, Xb, start, end, one_arg):
        cp = gp.coeffs_process
        factor = cp.cov_factor_ if cp._fit else gsum_amd.cov_factor(cp.scale0 ** 2, cp.df0)
        k = cp.kernel_ if cp._fit else kern
        K = k(Xa[rows], Xb)
        if one_arg:
            K[np.arange(len(rows)), rows] += 1e-5                             # the one-argument call's WhiteKernel diagonal
        rm = ratio(Xa[rows])[:, None] * ratio(Xb)
        return (ref(Xa[rows])[:, None] * ref(Xb)) * gsum_amd.geometric_sum(x=rm, start=start, end=end, excluded=[2]) * (factor * K)

    for fitted in (False, True):
        if fitted:
            y = gsum_amd.partials(rng.randn(n, r), ratio=ratio(X), ref=ref(X), orders=np.arange(r))
            gp.fit(X, y, orders=np.arange(r))
        for start, end in ((0, np.inf), (3, np.inf), (1, 4)):
            got = gp.cov(X, start=start, end=end)
            assert got.shape == (n, n)
            np.testing.assert_allclose(got[rows], want(X, X, start, end, True), rtol=1e-12)
            got = gp.cov(X, Xp, start=start, end=end)
            assert got.shape == (n, m)
            np.testing.assert_allclose(got[rows], want(X, Xp, start, end, False), rtol=1e-12)
    m0, s0 = gp.underlying_properties(X[:50], order=2, return_std=True)
    np.testing.assert_allclose(s0 ** 2, np.diag(gp.cov(X[:50], start=3)), rtol=1e-13)
    with pytest.raises(ValueError):
        gp.cov(X[:5], start=3, end=2)
    with pytest.raises(TypeError):
        gsum_amd.TruncationGP(ratio=0.5).ratio(X[:3], scale=2.0)            # a constant ratio takes `ratio=` only, like the reference's lambda
    assert np.all(gsum_amd.TruncationGP(ratio=0.5).ratio(X[:3], ratio=0.25) == 0.25)


@pytest.mark.parametrize("groups,size", [(3, 2), (2, 3), (3, 8), (4, 1)])
def test_batch_shares_cover_every_evaluation_once(lab, groups, size):
    """A call of several rounds hands its evaluations to the groups in equal shares (gs_lml_wave: R = ceil(n / (G B)) rounds, G R
    group-rounds of floor / ceil(n / (G R)) members).  Whatever the count -- fewer than groups, one more than a full round, a prime --
    every evaluation comes back once, in its position, bit-identical to a single evaluation; one member that is not positive
    definite does not disturb its neighbours."""
    ctx = lab
    from sklearn.gaussian_process.kernels import RBF
    n = 700
    rng = np.random.RandomState(11)
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([rng.randn(n, 3), np.ones((n, 1))], axis=1)
    ctx.set_inputs(X, Z)
    ells = np.linspace(0.15, 0.4, 53)
    descs = [gsum_amd.describe_kernel(RBF(float(e)), 1) for e in ells]
    old = {k: ctx.get_option(k) for k in ("wave_groups", "wave_size", "wave_min", "medium_path")}
    try:
        ctx.set_option("medium_path", 0)
        ctx.set_option("wave_min", 1000)
        G0, s0, i0 = ctx.lml_resident(descs, 1e-10)                       # one after the other
        assert np.all(i0 == 0) and len(set(s0.tolist())) == len(descs)
        ctx.set_option("wave_min", 3)
        ctx.set_option("wave_groups", groups)
        ctx.set_option("wave_size", size)
        for count in (3, 4, 5, 7, groups * size, groups * size + 1, 2 * groups * size - 1, 23, 53):
            G1, s1, i1 = ctx.lml_resident(descs[:count], 1e-10)
            np.testing.assert_array_equal(i1, i0[:count])
            np.testing.assert_array_equal(s1, s0[:count])
            np.testing.assert_array_equal(G1, G0[:count])
    finally:
        for k, v in old.items():
            ctx.set_option(k, v)


def test_mixed_calls_on_one_context_stay_bit_stable():
    """Since round 4 every schedule of a context runs on the same four streams (batch groups, the single factorisation's chain and
    auxiliary streams, the gradient sweep, the gradient batch's other evaluations).  Ten seconds of batch calls, single evaluations,
    gradients and operator-level factorisations in random order on ONE context (tools/gpu_mixed_soak.py; the long runs are
    profiles/r04_mixed_soak.log): every result bit-identical to the first of its kind, no chain time-out."""
    import json
    import sys
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_mixed_soak.py"), "10", "2048"], capture_output=True, text=True,
                         timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    rec = json.loads([line for line in run.stdout.splitlines() if line.startswith("{")][-1])
    assert rec["chain_aborts"] == 0 and rec["chain_probe"] == 1
    assert all(rec["calls"].get(k, 0) > 0 for k in ("batch", "single", "grad", "gradbatch", "factorize", "two")), rec


def test_launch_list_restatement_matches_the_library(lab):
    """tools/wave_plan.py restates the far launches of a batch call (members, M, K per launch) to price the PMC traffic of
    profiles/r04_gemm_pmc.json in algorithmic bytes.  Its flops must be the library's own record of a profiled call, launch for launch in
    total: same count, same sum."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from wave_plan import far_launches
    from sklearn.gaussian_process.kernels import RBF
    ctx = lab
    n = 3000
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, 3), np.ones((n, 1))], axis=1)
    ctx.set_inputs(X, Z)
    old = ctx.get_option("medium_path")
    try:
        ctx.set_option("medium_path", 0)
        descs = [gsum_amd.describe_kernel(RBF(0.2 + 0.001 * i), 1) for i in range(11)]
        ctx.lml_resident(descs, 1e-10)
        ctx.set_option("profile_gemm", 1)
        ctx.kernel_profile()
        ctx.lml_resident(descs, 1e-10)
        prof = ctx.kernel_profile()
    finally:
        ctx.set_option("profile_gemm", 0)
        ctx.set_option("medium_path", old)
    plan = far_launches(n, [4, 4, 3])
    assert prof["bulk_update"]["launches"] == len(plan)
    assert prof["bulk_update"]["flops"] == float(sum(x["flops"] for x in plan))


def test_gradient_batch_on_grouped_factorisations(ctx):
    """Round 4: gsum_lml_grad_batch factorises its kernels in ONE grouped call and runs the gradient stages on the factors the groups'
    workspaces hold.  Every member equals the single call bit for bit, in its position -- also when the batch is larger than one round of
    the groups (chunks) and when one member is not positive definite (its info says so, its neighbours are untouched)."""
    from gsum_amd.kernels import describe_gradient, describe_kernel
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C
    n = 1500
    rng = np.random.RandomState(5)
    X = np.sort(rng.rand(n, 1), axis=0) * 150.0
    Z = np.concatenate([rng.randn(n, 4), np.ones((n, 1))], axis=1)
    good = [C(1.0 + 0.1 * i) * RBF(0.2 + 0.02 * i) + WhiteKernel(1e-8, noise_level_bounds="fixed") for i in range(6)]
    bad = C(1.0) * RBF(4000.0) + WhiteKernel(1e-300, noise_level_bounds="fixed")          # numerically singular
    kernels = good[:3] + [bad] + good[3:]
    descs = [describe_kernel(k, 1) for k in kernels]
    prms = [describe_gradient(k, 1) for k in kernels]
    old = {k: ctx.get_option(k) for k in ("wave_groups", "wave_size")}
    try:
        singles = [ctx.lml_grad(d, p, X, Z, 0.0) for d, p in zip(descs, prms)]
        assert singles[3][2] > 0 and all(s[2] == 0 for i, s in enumerate(singles) if i != 3)
        for groups, size in ((3, 8), (2, 1), (1, 2)):                      # one round; chunks of two
            ctx.set_option("wave_groups", groups)
            ctx.set_option("wave_size", size)
            G, sld, info, tr, H = ctx.lml_grad_batch(descs, prms, X, Z, 0.0)
            for i, s in enumerate(singles):
                assert info[i] == s[2]
                if s[2] != 0:
                    continue
                np.testing.assert_array_equal(G[i], s[0])
                assert sld[i] == s[1]
                np.testing.assert_array_equal(tr[i], s[3])
                np.testing.assert_array_equal(H[i], s[4])
    finally:
        for k, v in old.items():
            ctx.set_option(k, v)
