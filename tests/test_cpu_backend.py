"""``backend='cpu'`` (gsum_amd/_cpu.py; BASELINE config 1, SURVEY.md 8(b) "New, additive"): the model classes on numpy / scipy /
scikit-learn behind the operator interface the HIP library replaces.

The checks ARE the GPU parity tests: the golden-vector test functions of tests/test_gpu_parity.py are called here with
``GSUM_BACKEND=cpu`` in the environment, so the same assertions against the reference's outputs run on the CPU suite (host logic of
every class: priors, Gram algebra, gradients, truncation layer, conditioning, Student-t) and on the GPU suite (the HIP kernels).
Nothing here touches a GPU, and the cpu backend never imports oracle/."""
import os
import sys

import numpy as np
import pytest

import gsum_amd
import test_gpu_parity as T          # collected as GPU tests in its own right; here only its functions are borrowed


@pytest.fixture()
def cpu_backend(monkeypatch):
    monkeypatch.setenv("GSUM_BACKEND", "cpu")

    def no_gpu(*a, **k):
        raise AssertionError("the cpu backend must not ask for a HIP context")
    monkeypatch.setattr("gsum_amd.conjugate.default_context", no_gpu)
    yield


def test_backend_selection(monkeypatch):
    from gsum_amd._cpu import CpuContext
    monkeypatch.delenv("GSUM_BACKEND", raising=False)
    assert gsum_amd.ConjugateGaussianProcess().backend == "hip"                      # the default is the device, never the CPU
    assert isinstance(gsum_amd.ConjugateGaussianProcess(backend="cpu")._context(), CpuContext)
    assert isinstance(gsum_amd.TruncationGP(backend="cpu").coeffs_process._context(), CpuContext)
    monkeypatch.setenv("GSUM_BACKEND", "cpu")
    assert gsum_amd.TruncationTP().coeffs_process.backend == "cpu"
    assert gsum_amd.ConjugateGaussianProcess(backend="hip").backend == "hip"          # the argument wins over the environment
    with pytest.raises(ValueError):
        gsum_amd.ConjugateGaussianProcess(backend="cuda")
    src = open(os.path.join(os.path.dirname(gsum_amd.__file__), "_cpu.py")).read()
    assert "import oracle" not in src and "from oracle" not in src


def test_config1_fit_on_128_points(cpu_backend):
    """BASELINE configs[0]: ConjugateGaussianProcess.fit on n = 128 1-D points, 4 orders, RBF(0.2), CPU path -- the S1 workload
    against the reference's own outputs (tests/golden/s1_plumbing.json).  cond(K) = 5.6e11; the CPU backend issues the reference's
    own LAPACK / scikit-learn calls, the remaining difference is the Gram-matrix form of the scalar algebra (cond x eps)."""
    import test_gpu_round3 as T3
    T3.test_config1_s1_workload_through_the_small_path()
    from conftest import load_golden
    from sklearn.gaussian_process.kernels import RBF
    d = load_golden("s1_plumbing.json")
    X, y = np.linspace(0, 1, d["n"])[:, None], np.array(d["y"])
    gp = gsum_amd.ConjugateGaussianProcess(kernel=RBF(d["length_scale"]), center=0, disp=0, df=1, scale=1, optimizer=None)
    gp.fit(X, gsum_amd.coefficients(y, ratio=d["ratio"], ref=d["ref"], orders=np.arange(d["r"])))
    assert gp.df_ == d["df"] and gp._context().calls["potrf"] >= 1


def test_cgp_golden(cpu_backend, small_cases):
    T.test_cgp_lml_fit_predict_golden(small_cases)


def test_trunc_golden(cpu_backend, small_cases):
    T.test_trunc_lml_golden(small_cases)
    T.test_trunc_lml_array_ratio_ref(small_cases)
    T.test_nonpd_behaviour(small_cases)


def test_truncation_predict_golden(cpu_backend):
    T.test_truncation_predict_golden()
    T.test_truncation_predict_all_kinds_golden()


def test_student_golden(cpu_backend):
    T.test_student_process_golden()


def test_gradient_golden(cpu_backend):
    T.test_lml_gradient_golden()


def test_underlying_and_sample_y_golden(cpu_backend):
    T.test_underlying_properties_golden()
    T.test_sample_y_default_path_reproduces_the_reference_draws()


@pytest.mark.parametrize("name", T.REFERENCE_TEST_KERNELS)
@pytest.mark.parametrize("decomposition", ["cholesky", "eig"])
def test_reference_interpolation_test(cpu_backend, name, decomposition):
    T.test_interpolation_property(name, decomposition)


def test_notebook_grid_and_optimizer(cpu_backend, notebook_grid):
    T.test_notebook_grid_known_answer(notebook_grid)
    T.test_fit_with_optimizer_reaches_the_grid_optimum(notebook_grid)


def test_large_known_answer_n512(cpu_backend, large_lml):
    T.test_large_known_answers_uniform_grid(large_lml, 0)
    T.test_large_known_answers_gp_drawn(0)


def test_tree_kernels_golden(cpu_backend):
    T.test_tree_kernels_golden()


def test_predict_var_contract_on_the_cpu_backend():
    """gsum_predict_var (SURVEY.md 8b's name for the predictive pieces, models.py:822-836) on the cpu backend's operator interface: the same
    contract as the HIP entry point -- sums of squares always, V^T W only for right-hand sides a forward_gram left on the factor."""
    import numpy as np
    from sklearn.gaussian_process.kernels import Matern, WhiteKernel
    from gsum_amd._cpu import CpuContext
    from gsum_amd._lib import GSUM_MAX_RHS
    rng = np.random.RandomState(3)
    X, Xs, rhs = rng.rand(40, 2) * 4, rng.rand(9, 2) * 4, rng.randn(40, 3)
    desc = gsum_amd.describe_kernel(Matern([0.7, 1.3], nu=2.5) + WhiteKernel(1e-6), 2)
    ctx = CpuContext()
    L, info = ctx.factorize(desc, X, diag_add=1e-10)
    assert info == 0
    css0, none = ctx.predict_var(L, desc, X, Xs)
    assert none is None
    with pytest.raises(ValueError, match="solved"):
        ctx.predict_var(L, desc, X, Xs, want_vtw=True)
    ctx.forward_gram(L, rhs)
    css, vtw = ctx.predict_var(L, desc, X, Xs, want_vtw=True)
    want = ctx.predict_terms(L, desc, X, Xs, rhs=rhs)
    np.testing.assert_allclose(css, want[0], rtol=1e-13)
    assert np.array_equal(css, css0) and vtw.shape == (9, GSUM_MAX_RHS) and not vtw[:, 3:].any()
    np.testing.assert_allclose(vtw[:, :3], want[1], rtol=1e-12, atol=1e-13)


def test_eig_mode_golden(cpu_backend):
    T.test_eig_mode_golden()


@pytest.mark.parametrize("r", [16, 33])
def test_more_curves_than_one_device_call_takes(cpu_backend, r):
    T.test_more_curves_than_one_device_call_takes(r)


def test_an_empty_batch_is_no_work_on_the_cpu_backend():
    """The same contract as the HIP entry points (tests/test_gpu_round5.py): empty outputs of the right shapes, empty surfaces."""
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel
    from gsum_amd._cpu import CpuContext
    rng = np.random.RandomState(0)
    X = np.sort(rng.rand(30))[:, None] * 9
    Z = np.concatenate([rng.randn(30, 2), np.ones((30, 1))], axis=1)
    ctx = CpuContext()
    G, sld, info = ctx.lml_batch([], X, Z, 1e-10)
    assert G.shape == (0, 3, 3) and sld.shape == (0,) and info.shape == (0,)
    gp = gsum_amd.TruncationGP(kernel=RBF(0.5) + WhiteKernel(1e-4), ratio=0.5, ref=1.0, optimizer=None, center=0, disp=0, df=1, scale=1,
                               backend="cpu")
    gp.fit(X, gsum_amd.partials(Z[:, :2], ratio=0.5, ref=1.0, orders=np.arange(2)), orders=np.arange(2))
    assert gp.log_marginal_likelihood_grid([], [0.4, 0.6], mode="full").shape == (2, 0)
    assert gp.log_marginal_likelihood_grid([gp.coeffs_process.kernel_.theta], [], mode="reuse").shape == (0, 1)
