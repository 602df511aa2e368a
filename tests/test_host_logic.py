"""CPU tests of the host side of gsum_amd: series arithmetic, kernel-tree flattening, the Gram-matrix
algebra that replaces the reference's cho_solve calls (checked against the golden vectors with a
numpy-computed Gram matrix), the C-ABI export table, and the option / error behaviour that does not need
a GPU."""
import os
import subprocess
import re

import numpy as np
import pytest
from scipy.linalg import solve_triangular

import gsum_amd
from conftest import ROOT, make_kernel, prior_kwargs
from gsum_amd.conjugate import lml_from_gram, posterior_from_gram


def test_series_helpers_match_reference(small_cases):
    h = small_cases["helpers"]
    y, ratio, ref = np.array(h["y"]), np.array(h["ratio"]), np.array(h["ref"])
    orders = np.array(h["orders"])
    np.testing.assert_array_equal(gsum_amd.coefficients(y, ratio, ref, orders), np.array(h["coefficients_arr"]))
    np.testing.assert_array_equal(gsum_amd.coefficients(y, 0.4, 2.0), np.array(h["coefficients_scalar"]))
    np.testing.assert_array_equal(gsum_amd.partials(y, ratio, ref, orders), np.array(h["partials_arr"]))
    np.testing.assert_array_equal(gsum_amd.partials(y, 0.4, 2.0), np.array(h["partials_scalar"]))
    x = np.array(h["geo_x"])
    np.testing.assert_array_equal(gsum_amd.geometric_sum(x, 0, np.inf), np.array(h["geo_0_inf"]))
    np.testing.assert_array_equal(gsum_amd.geometric_sum(x, 2, 5), np.array(h["geo_2_5"]))
    np.testing.assert_array_equal(gsum_amd.geometric_sum(x, 1, np.inf, excluded=[2, 7]), np.array(h["geo_1_inf_excl"]))
    np.testing.assert_array_equal(gsum_amd.geometric_sum(x, 3, 6, excluded=4), np.array(h["geo_3_6_excl"]))
    with pytest.raises(ValueError):
        gsum_amd.geometric_sum(x, 3, 2)
    with pytest.raises(ValueError):
        gsum_amd.coefficients(y[:, 0], 0.5)
    with pytest.raises(ValueError):
        gsum_amd.coefficients(y, 0.5, orders=[0, 1])


def _gram(kern, theta, X, y, nugget=1e-10):
    k = kern.clone_with_theta(theta) if theta is not None else kern
    R = k(X)
    R[np.diag_indices_from(R)] += nugget
    L = np.linalg.cholesky(R)
    if y.ndim == 1:
        y = y[:, None]
    Z = np.concatenate([y, np.ones((len(X), 1))], axis=1)
    W = solve_triangular(L, Z, lower=True)
    return W.T @ W, np.log(np.diag(L)).sum()


def test_gram_algebra_reproduces_reference_lml_and_posterior(small_cases):
    """lml_from_gram / posterior_from_gram against reference outputs, all prior regimes."""
    for case in small_cases["cgp"]:
        kern = make_kernel(case["kernel"])
        pk = prior_kwargs(case["prior"])
        gp = gsum_amd.ConjugateGaussianProcess(kernel=kern, optimizer=None, **pk)
        X, y = np.array(case["X"]), np.array(case["y"])
        G, sld = _gram(kern, np.array(case["theta"]), X, y)
        lml, _ = lml_from_gram(G, sld, len(X), gp.center0, gp.disp0, gp.df0, gp.scale0)
        assert lml == pytest.approx(case["lml_theta"], rel=1e-11)
        G1, sld1 = _gram(kern, np.array(case["theta"]), X, y[:, 0])
        lml1, _ = lml_from_gram(G1, sld1, len(X), gp.center0, gp.disp0, gp.df0, gp.scale0)
        assert lml1 == pytest.approx(case["lml_1col"], rel=1e-11)
        G, sld = _gram(kern, None, X, y)
        lml, post = lml_from_gram(G, sld, len(X), gp.center0, gp.disp0, gp.df0, gp.scale0)
        g = case["fit"]
        assert lml == pytest.approx(g["lml"], rel=1e-11)
        # center/disp are ratios of R^-1 bilinear forms: on the randomly spaced (ill-conditioned) golden
        # inputs two valid evaluation orders differ at ~cond * eps
        np.testing.assert_allclose(post["center"], g["center"], rtol=1e-7, atol=1e-12)
        np.testing.assert_allclose(post["disp"], g["disp"], rtol=1e-7, atol=1e-15)
        assert post["df"] == g["df"]
        assert np.sqrt(post["scale_sq"]) == pytest.approx(g["scale"], rel=1e-11)
        assert post["cov_factor"] == pytest.approx(g["cov_factor"], rel=1e-11)


def test_batched_gram_algebra_equals_scalar_version():
    from gsum_amd.conjugate import lml_from_gram_batch
    rng = np.random.RandomState(3)
    Gs, ss = [], []
    for _ in range(7):
        W = rng.randn(40, 5)
        W[:, -1] = np.abs(W[:, -1])
        Gs.append(W.T @ W)
        ss.append(rng.randn())
    for pri in ((0, 0, 1, 1), (0.3, 0, 3, 1.5), (0.2, 2.0, 1, 1), (-0.1, 0.7, np.inf, 1.2), (-0.4, 0, np.inf, 0.8)):
        want = [lml_from_gram(G, s_, 40, *pri)[0] for G, s_ in zip(Gs, ss)]
        got = lml_from_gram_batch(np.array(Gs), np.array(ss), 40, *pri)
        np.testing.assert_allclose(got, want, rtol=1e-13)


def test_posterior_rejects_vector_priors():
    with pytest.raises(ValueError):
        posterior_from_gram(np.eye(3), 10, [0.0, 1.0], 0, 1, 1)


def test_describe_kernel_flattening():
    from sklearn.gaussian_process.kernels import RBF, Matern, WhiteKernel, ConstantKernel as C, RationalQuadratic
    d = gsum_amd.describe_kernel(C(1.7) * RBF([0.3, 0.4]) + WhiteKernel(1e-3) + C(0.5), 2)
    assert (d.family, d.anisotropic, d.amplitude, d.additive_const, d.white_noise) == (0, 1, 1.7, 0.5, 1e-3)
    assert list(d.length_scale)[:2] == [0.3, 0.4]
    d = gsum_amd.describe_kernel(Matern(0.3, nu=2.5), 3)
    assert (d.family, d.anisotropic, d.length_scale[0], d.amplitude) == (1, 0, 0.3, 1.0)
    assert gsum_amd.describe_kernel(Matern(0.3, nu=1.5), 1).family == 2
    assert gsum_amd.describe_kernel(Matern(0.3, nu=0.5), 1).family == 3
    d = gsum_amd.describe_kernel(C(2.0) * (C(3.0) * RBF(1.0)), 1)
    assert d.amplitude == 6.0
    # theta ordering / log-parameters stay with scikit-learn
    k = (C(1.0) * RBF(1.0) + WhiteKernel(1e-2, noise_level_bounds="fixed")).clone_with_theta(np.log([2.0, 0.5]))
    d = gsum_amd.describe_kernel(k, 1)
    assert d.amplitude == pytest.approx(2.0) and d.length_scale[0] == pytest.approx(0.5)
    # outside the flattened family: a postfix program over the tree (round 4), in scikit-learn's evaluation and theta order
    from sklearn.gaussian_process.kernels import DotProduct, ExpSineSquared
    from gsum_amd._lib import OP_ADD, OP_MUL, OP_LEAF, OP_CONST, OP_WHITE
    t = gsum_amd.describe_kernel(RBF(1.0) + RBF(2.0), 1)
    assert t.is_tree and list(t.op)[:t.n_ops] == [OP_LEAF, OP_LEAF + 1, OP_ADD] and t.n_leaves == 2
    assert (t.leaf[0].length_scale[0], t.leaf[1].length_scale[0]) == (1.0, 2.0)
    t = gsum_amd.describe_kernel(C(2.0) * RBF(0.5) + C(0.5) * Matern(1.5, nu=2.5) + WhiteKernel(1e-3), 1)
    assert list(t.op)[:t.n_ops] == [OP_CONST, OP_LEAF, OP_MUL, OP_CONST + 1, OP_LEAF + 1, OP_MUL, OP_ADD, OP_WHITE + 2, OP_ADD]
    assert (t.cval[0], t.cval[1], t.cval[2]) == (2.0, 0.5, 1e-3) and t.one_arg_diagonal() == (2.0 * 1.0 + 0.5 * 1.0) + 1e-3
    assert t.without_white().one_arg_diagonal() == 2.5 and t.plus_constant(0.25).one_arg_diagonal() == t.one_arg_diagonal() + 0.25
    rq = gsum_amd.describe_kernel(RationalQuadratic(length_scale=0.7, alpha=1.3), 1)
    assert rq.is_tree and (rq.leaf[0].family, rq.leaf[0].alpha, rq.leaf[0].length_scale[0]) == (4, 1.3, 0.7)
    assert gsum_amd.describe_kernel(RBF(1.0) * RBF(2.0), 1).is_tree
    k = RationalQuadratic(length_scale=0.7, alpha=1.3) * RBF([1.0, 2.0]) + C(0.3)
    th = k.theta + 0.1
    assert bytes(gsum_amd.describe_thetas(k, [th], 2)[0]) == bytes(gsum_amd.describe_kernel(k.clone_with_theta(th), 2))
    from gsum_amd.kernels import describe_gradient
    from gsum_amd._lib import GradParam
    assert [(g.code, g.dim) for g in describe_gradient(k, 2)] == [(GradParam.TREE_ALPHA, 0), (GradParam.TREE_LENGTH_ISO, 0),
                                                                 (GradParam.TREE_LENGTH_DIM, 16), (GradParam.TREE_LENGTH_DIM, 17), (GradParam.TREE_CONST, 0)]
    # round 5: ExpSineSquared (second parameter = periodicity, theta order length_scale, periodicity), Matern(nu = inf), Exponentiation
    es = gsum_amd.describe_kernel(C(2.0) * ExpSineSquared(length_scale=1.1, periodicity=3.0) ** 2 + WhiteKernel(0.5), 1)
    assert es.is_tree and (es.leaf[0].family, es.leaf[0].alpha, es.leaf[0].length_scale[0]) == (5, 3.0, 1.1)
    assert [es.op[i] for i in range(es.n_ops)] == [32 + 0, 16 + 0, 128 + 1, 2, 64 + 2, 1] and es.cval[1] == 2.0
    assert es.one_arg_diagonal() == 2.0 * 1.0 ** 2 + 0.5 and es.without_white().one_arg_diagonal() == 2.0
    assert es.plus_constant(0.25).one_arg_diagonal() == 2.75 and es.plus_constant(0.25).cval[3] == 0.25       # (a slot no operand uses)
    k2 = ExpSineSquared(length_scale=1.1, periodicity=3.0) * Matern([1.0, 2.0], nu=np.inf)
    assert gsum_amd.describe_kernel(k2, 2).leaf[1].family == 6
    assert [(g.code, g.dim) for g in describe_gradient(k2, 2)] == [(GradParam.TREE_LENGTH_ISO, 0), (GradParam.TREE_ALPHA, 0),
                                                                  (GradParam.TREE_LENGTH_DIM, 16), (GradParam.TREE_LENGTH_DIM, 17)]
    th2 = k2.theta - 0.2
    assert bytes(gsum_amd.describe_thetas(k2, [th2], 2)[0]) == bytes(gsum_amd.describe_kernel(k2.clone_with_theta(th2), 2))
    dp = gsum_amd.describe_kernel(C(0.5) * DotProduct(sigma_0=0.7) ** 2 + RBF(1.0), 2)           # the one leaf whose diagonal is not 1
    assert dp.leaf[0].family == 7 and dp.leaf[0].length_scale[0] == 0.7
    Xd = np.array([[1.0, 2.0], [0.5, -1.0]])
    np.testing.assert_array_equal(dp.one_arg_diagonal(Xd), (C(0.5) * DotProduct(sigma_0=0.7) ** 2 + RBF(1.0)).diag(Xd))
    with pytest.raises(ValueError):
        dp.one_arg_diagonal()
    cw = gsum_amd.describe_kernel(C(0.5) + WhiteKernel(0.1), 1)                  # no leaf at all: c 1 1^T + w I is a legal kernel
    assert cw.is_tree and cw.n_leaves == 0 and cw.one_arg_diagonal() == 0.6 and gsum_amd.describe_kernel(WhiteKernel(1.0), 1).n_ops == 1
    for bad in (Matern(1.0, nu=3.5),
                RBF(1.0) + RBF(2.0) + RBF(3.0) + RBF(4.0) + RBF(5.0)):
        with pytest.raises(NotImplementedError):
            gsum_amd.describe_kernel(bad, 1)
    with pytest.raises(ValueError):
        gsum_amd.describe_kernel(RBF([1.0, 2.0]), 3)
    with pytest.raises(ValueError):
        gsum_amd.describe_kernel(RBF(1.0), 9)


def test_library_exports_every_declared_symbol():
    """include/gsum_hip.h is the contract: every function it declares must be exported and bound."""
    from gsum_amd import _lib
    header = open(os.path.join(ROOT, "include", "gsum_hip.h")).read()
    declared = set(re.findall(r"\b(gsum_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    lib = _lib.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in gsum_hip.h but not exported"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    import ctypes
    leaf = 8 + 8 * _lib.GSUM_MAX_D + 8
    assert ctypes.sizeof(_lib.KernelLeaf) == leaf
    assert ctypes.sizeof(_lib.KernelDesc) == (8 + 8 * _lib.GSUM_MAX_D + 24) + 8 + 4 * _lib.GSUM_MAX_OPS + 8 * _lib.GSUM_MAX_OPS + leaf * _lib.GSUM_MAX_LEAVES
    # the C compiler agrees with the ctypes mirror (struct layout is part of the contract)
    import shutil, tempfile
    if shutil.which("gcc"):
        with tempfile.TemporaryDirectory() as tmp:
            src = os.path.join(tmp, "sz.c")
            open(src, "w").write('#include "gsum_hip.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(void){printf("%zu %zu %zu %zu", sizeof(gsum_kernel_desc), '
                                 'sizeof(gsum_kernel_leaf), offsetof(gsum_kernel_desc, n_ops), offsetof(gsum_kernel_desc, leaf));return 0;}')
            subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), src, "-o", os.path.join(tmp, "sz")], check=True)
            got = subprocess.run([os.path.join(tmp, "sz")], capture_output=True, text=True, check=True).stdout.split()
        assert [int(v) for v in got] == [ctypes.sizeof(_lib.KernelDesc), leaf, _lib.KernelDesc.n_ops.offset, _lib.KernelDesc.leaf.offset]
    # ... and NOTHING else: the product library is the contract (33 single-GPU entry points + the 14 of the device group of round 5, a
    # header a maintainer can read in one sitting);
    # diagnostics, probes and schedule switches are the lab build's (include/gsum_hip_debug.h, libgsum_hip_lab.so)
    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith("gsum_")}
    assert exported(_lib.LIB_PATH) == declared and len(declared) <= 47
    assert len(header.splitlines()) < 260
    debug = open(os.path.join(ROOT, "include", "gsum_hip_debug.h")).read()
    lab_declared = set(re.findall(r"\b(gsum_[a-z0-9_]+)\s*\(", debug)) - declared
    assert lab_declared == set(_lib.LAB_PROTOTYPES), lab_declared ^ set(_lib.LAB_PROTOTYPES)
    assert exported(_lib.LAB_LIB_PATH) == declared | lab_declared
    for name in ("gsum_debug", "gsum_probe", "gsum_bench"):
        assert name not in header


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gsum_amd import _lib
    with pytest.raises(RuntimeError, match="gsum_init"):
        _lib.HipContext(0)
    from sklearn.gaussian_process.kernels import RBF
    gp = gsum_amd.ConjugateGaussianProcess(kernel=RBF(1.0), optimizer=None)
    with pytest.raises(RuntimeError):
        gp.fit(np.arange(5.0)[:, None], np.arange(5.0))


def test_missing_library_is_reported(tmp_path):
    from gsum_amd import _lib
    with pytest.raises(RuntimeError, match="not built"):
        _lib.load_library(str(tmp_path / "nope.so"))


def test_constructor_and_argument_errors():
    from sklearn.gaussian_process.kernels import RBF
    gp = gsum_amd.ConjugateGaussianProcess(kernel=RBF(1.0), sd=2.0)
    assert gp.df0 == np.inf and gp.scale0 == 2.0
    assert gp.center0.shape == (1,) and gp.disp0.shape == (1, 1)
    with pytest.raises(NotImplementedError):
        gsum_amd.ConjugateGaussianProcess(basis=lambda X: X)
    with pytest.raises(RuntimeError):
        gp._fit = True
        gp.predict(np.zeros((2, 1)), return_std=True, return_cov=True)
    gp._fit = False
    with pytest.raises(ValueError):
        gsum_amd.ConjugateGaussianProcess(decomposition="lu").log_marginal_likelihood(np.array([0.0]), X=np.zeros((2, 1)), y=np.zeros(2))
    assert gsum_amd.ConjugateGaussianProcess(decomposition="eig")._check_decomposition() is None          # accepted since round 5 (models.py:713-717)
    with pytest.raises(ValueError):
        gsum_amd.ConjugateGaussianProcess().cov(np.zeros((2, 1)))     # df0 = 1 <= 2: covariance does not exist
    t = gsum_amd.TruncationGP(kernel=RBF(1.0), ratio=lambda X: np.ones((len(X), 1, 1)))
    with pytest.raises(ValueError):
        t.fit(np.zeros((3, 1)), np.zeros((3, 2)), orders=np.arange(2))


def test_gradient_parameter_map_follows_sklearn_theta_order():
    """describe_gradient lists one entry per component of kernel.theta, in scikit-learn's order (leaves left to
    right, free hyperparameters only): SURVEY.md quirk Q10."""
    from sklearn.gaussian_process.kernels import RBF, Matern, WhiteKernel, ConstantKernel as C
    from gsum_amd.kernels import describe_gradient
    from gsum_amd._lib import GradParam as P
    k = C(1.3) * Matern([0.5, 0.7], nu=2.5) + WhiteKernel(0.01) + C(0.2)
    got = [(g.code, g.dim, g.weight) for g in describe_gradient(k, 2)]
    assert got == [(P.AMPLITUDE, 0, 0.0), (P.LENGTH_DIM, 0, 0.0), (P.LENGTH_DIM, 1, 0.0), (P.WHITE, 0, 0.01), (P.ADDITIVE, 0, 0.2)]
    assert len(got) == len(k.theta)
    k = RBF(0.5) + WhiteKernel(1e-6, noise_level_bounds="fixed")
    assert [(g.code, g.dim) for g in describe_gradient(k, 3)] == [(P.LENGTH_ISO, 0)]
    k = WhiteKernel(0.1) + C(2.0, constant_value_bounds="fixed") * RBF(1.0)
    assert [(g.code, g.weight) for g in describe_gradient(k, 1)] == [(P.WHITE, 0.1), (P.LENGTH_ISO, 0.0)]
    assert describe_gradient(C(1.0, constant_value_bounds="fixed") * RBF(1.0, length_scale_bounds="fixed"), 1) == []
    assert [(g.code, g.dim) for g in describe_gradient(RBF(1.0) + RBF(2.0), 1)] == [(P.TREE_LENGTH_ISO, 0), (P.TREE_LENGTH_ISO, 16)]     # a tree
    from sklearn.gaussian_process.kernels import DotProduct, PairwiseKernel
    assert [(g.code, g.dim) for g in describe_gradient(DotProduct() + RBF(2.0), 1)] == [(P.TREE_LENGTH_ISO, 0), (P.TREE_LENGTH_ISO, 16)]   # (sigma_0 rides in length_scale[0])
    with pytest.raises(NotImplementedError):
        describe_gradient(PairwiseKernel() + RBF(2.0), 1)


def test_series_scale_struct_and_student_host_algebra():
    """SeriesScale mirrors helpers.geometric_sum's argument checks; the Student likelihood from a Gram matrix equals
    the oracle's dense computation (no GPU involved: G is built with numpy here)."""
    from gsum_amd._lib import SeriesScale
    from gsum_amd.conjugate import student_lml_from_gram
    from sklearn.gaussian_process.kernels import RBF
    import sys
    sys.path.insert(0, ROOT)
    from oracle import gsum_oracle as orc               # checker only
    sc = SeriesScale.make(2, np.inf, excluded=[3, 5], factor=1.5)
    assert (sc.start, sc.end, sc.n_excluded, list(sc.excluded)[:2], sc.factor) == (2, -1, 2, [3, 5], 1.5)
    with pytest.raises(ValueError):
        SeriesScale.make(3, 2)
    rng = np.random.RandomState(0)
    X = np.linspace(0, 1, 12)[:, None]
    y = rng.randn(12, 3)
    kern = RBF(0.1)                                     # well conditioned: this checks algebra, not rounding
    R = kern(X) + 1e-10 * np.eye(12)
    Z = np.c_[y, np.ones(12)]
    G = Z.T @ np.linalg.solve(R, Z)
    sld = np.log(np.diag(np.linalg.cholesky(R))).sum()
    for disp in (0, 1.7):
        got, _ = student_lml_from_gram(G, sld, 12, np.array([0.2]), np.array([[disp]]), 3.0, 1.1)
        want = orc.csp_lml(kern, None, X, y, center=0.2, disp=disp, df=3.0, scale=1.1)
        assert got == pytest.approx(want, rel=1e-9)


def _classmethod_case_inputs(g):
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel as C
    X, y = np.array(g["X"]), np.array(g["y"])
    kern = C(g["kernel"]["amplitude"]) * RBF(g["kernel"]["length_scale"])
    R, dR = kern(X, eval_gradient=True)
    R[np.diag_indices_from(R)] += g["nugget"]
    return X, y, np.linalg.cholesky(R), dR, R


def test_reference_classmethod_surface_against_reference():
    """SURVEY.md 8(b) 'Python surface to preserve', name by name: the compute_* / solve_sqrt / num_y / avg_y classmethods
    with the reference's signatures (models.py:170-503, 601-628), on a host factor, against outputs of the reference
    itself (tests/golden/classmethods.json)."""
    from conftest import load_golden
    g = load_golden("classmethods.json")
    X, y, chol, dR, R = _classmethod_case_inputs(g)
    n = len(X)
    for cls in (gsum_amd.ConjugateGaussianProcess, gsum_amd.ConjugateStudentProcess):
        for name in ("compute_center", "compute_disp", "compute_df", "compute_scale_sq", "compute_cov_factor",
                     "solve_sqrt", "num_y", "avg_y"):
            assert callable(getattr(cls, name)), name
    cgp = gsum_amd.ConjugateGaussianProcess
    for case in g["cases"]:
        basis = np.ones((n, 1)) if case["basis_cols"] == 1 else np.concatenate([np.ones((n, 1)), X], axis=1)
        center0, disp0 = np.array(case["center0"]), np.array(case["disp0"])
        df0 = np.inf if case["df0"] == "inf" else case["df0"]
        scale0 = case["scale0"]
        c, dc = cgp.compute_center(y, chol, basis, center0, disp0, 'cholesky', eval_gradient=True, dR=dR)
        V, dV = cgp.compute_disp(y, chol, basis, disp0, 'cholesky', eval_gradient=True, dR=dR)
        df, ddf = cgp.compute_df(y, df0, eval_gradient=True, dR=dR)
        s2, ds2 = cgp.compute_scale_sq(y, chol, basis, center0, disp0, df0, scale0, 'cholesky', eval_gradient=True, dR=dR)
        for got, key in ((c, "center"), (dc, "d_center"), (V, "disp"), (dV, "d_disp"), (ddf, "d_df"), (ds2, "d_scale_sq")):
            want = np.array(case[key])
            assert np.shape(got) == want.shape, key
            np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(want).max()), err_msg=key)
        assert df == case["df"]
        assert s2 == pytest.approx(case["scale_sq"], rel=1e-10)
        assert cgp.compute_cov_factor(s2, df) == pytest.approx(case["cov_factor"], rel=1e-10)
        # value-only calls, 1-D y, keyword form as the reference's own callers use them (models.py:1001-1002)
        np.testing.assert_allclose(cgp.compute_center(y=y[:, 0], sqrt_R=chol, basis=basis, center0=center0, disp0=disp0,
                                                      decomposition='cholesky'), case["center_1d_y"], rtol=1e-9)
        assert cgp.compute_scale_sq(y=y[:, 0], sqrt_R=chol, basis=basis, center0=center0, disp0=disp0, df0=df0,
                                    scale0=scale0, decomposition='cholesky') == pytest.approx(case["scale_sq_1d_y"], rel=1e-10)
        with pytest.raises(ValueError):
            cgp.compute_disp(y, chol, basis, disp0, 'cholesky', eval_gradient=True)       # dR missing (models.py:263)
    ss = g["solve_sqrt"]
    B = np.array(ss["B"])
    w, Q = np.linalg.eigh(R)
    np.testing.assert_allclose(cgp.solve_sqrt(chol, B, 'cholesky'), ss["chol"], rtol=1e-9)
    np.testing.assert_allclose(cgp.solve_sqrt(chol, B[:, 0], 'cholesky'), ss["vec"], rtol=1e-9)
    np.testing.assert_allclose(cgp.solve_sqrt((w, Q), B, 'eig'), ss["eig_tuple"], rtol=1e-6)
    np.testing.assert_allclose(cgp.solve_sqrt(Q * np.sqrt(w), B, 'eig'), ss["eig_sqrt"], rtol=1e-6)
    with pytest.raises(ValueError):
        cgp.solve_sqrt(chol, B, 'lu')                                                     # models.py:477
    assert [cgp.num_y(y), cgp.num_y(y[:, 0])] == g["num_y"]
    np.testing.assert_allclose(cgp.avg_y(y), g["avg_y"], rtol=1e-14)


def test_constant_only_product_terms_are_additive():
    """A Product with no stationary factor is an additive constant (advisor finding, round 1): C(2) * C(3) adds 6
    everywhere and must not overwrite the amplitude of the stationary term."""
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C
    d = gsum_amd.describe_kernel(RBF(0.3) + C(2.0) * C(3.0), 1)
    assert (d.amplitude, d.additive_const) == (1.0, 6.0)
    d = gsum_amd.describe_kernel(C(2.0) * C(3.0) + C(1.5) * RBF(0.3) + WhiteKernel(0.1), 1)
    assert (d.amplitude, d.additive_const, d.white_noise) == (1.5, 6.0, 0.1)
    from gsum_amd.kernels import describe_gradient
    from gsum_amd._lib import GradParam
    k = C(1.5) * RBF(0.3) + C(2.0) * C(3.0, constant_value_bounds="fixed")
    gp = describe_gradient(k, 1)
    assert [p.code for p in gp] == [GradParam.AMPLITUDE, GradParam.LENGTH_ISO, GradParam.ADDITIVE]
    assert gp[2].weight == 6.0
    X = np.linspace(0, 1, 5)[:, None]
    K, dK = k(X, eval_gradient=True)
    np.testing.assert_allclose(dK[:, :, 2], 6.0)            # d (c1 c2) / d log c1 = c1 c2 everywhere
    cc = gsum_amd.describe_kernel(C(2.0) * C(3.0), 1)         # (a kernel without a leaf is a tree of constants since late round 5)
    assert cc.is_tree and cc.n_leaves == 0 and cc.one_arg_diagonal() == 6.0


def test_importing_the_package_leaves_the_environment_alone():
    """Nothing in the package touches GPU_MAX_HW_QUEUES any more (rounds 1-3 asked applications to raise it for the
    one-stream-per-evaluation batch; the grouped batch schedule of round 4 runs on three streams)."""
    import subprocess
    import sys
    code = ("import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import gsum_amd; "
            "assert 'GPU_MAX_HW_QUEUES' not in os.environ; assert not hasattr(gsum_amd, 'configure_runtime')")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
    for name in ("bench.py", os.path.join("tests", "conftest.py"), os.path.join("gsum_amd", "_lib.py")):
        src = open(os.path.join(ROOT, name)).read()
        assert 'setdefault("GPU_MAX_HW_QUEUES"' not in src and 'environ["GPU_MAX_HW_QUEUES"] =' not in src, name


def test_bench_launch_command_is_the_contracts_and_needs_no_gpu():
    """`python bench.py --gpus N` starts its N ranks through exactly the launcher line the driver uses (one process per GPU,
    rendezvous on 127.0.0.1); building that command is host logic and touches no GPU (importing bench does not import torch)."""
    import importlib
    import sys as _sys
    had_torch = "torch" in _sys.modules
    bench = importlib.import_module("bench")
    cmd = bench.launch_command(8, 29555, ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert cmd[:3] == [_sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert had_torch or "torch" not in _sys.modules
    assert bench.PARITY_BOUND == 1e-10
    assert bench.golden_lml(8192, 6) == -212339.01508125057 and bench.golden_lml(2048, 4) == -41135.57871678863
    assert bench.golden_lml(8192, 5) is None


def test_kernel_register_budgets():
    """The schedules count on what a wave of each kernel allocates: six bulk waves (<= 72 VGPRs each, allocated in eights) leave
    room on a SIMD for ONE chain / panel wave (<= 224) as soon as a bulk workgroup retires; with 80 the panel of the rows below
    the window took 270 us instead of 80 beside the trailing update (found in round 3 after an innocent-looking edit).  Compiles
    the device code once with -Rpass-analysis=kernel-resource-usage (hipcc cross-compiles without a GPU; ~30 s)."""
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    src = os.path.join(ROOT, "gsum_amd", "csrc", "gsum_capi.hip")
    with tempfile.TemporaryDirectory() as tmp:
        res = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c",
                              "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "gsum_amd", "csrc"),
                              "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(tmp, "dev.o"), src],
                             capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    usage, name = {}, None
    for line in res.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
        for key in ("VGPRs", "AGPRs", "ScratchSize \\[bytes/lane\\]"):
            m = re.search(r"remark:\s+" + key + r": (\d+)", line)
            if m and name:
                usage[name][key[:7]] = int(m.group(1))

    def find(prefix):
        hits = [v for k, v in usage.items() if prefix in k]
        assert hits, prefix
        return hits[0]

    bulk = find("k_gemm_ld3ILi2E")
    assert bulk["VGPRs"] <= 72 and bulk["Scratch"] == 0, bulk
    for grouped in ("11k_gemm_ld3g", "11k_gemm_ld3n"):                       # the batch schedule's launches: the same tile, the same budget
        g = find(grouped)
        assert g["VGPRs"] <= 72 and g["Scratch"] == 0, (grouped, g)
    # bulk waves allocate 64 registers: six leave a SIMD 128, every retiring bulk workgroup 128 more -- a panel wave of up to 256 starts
    # where ONE bulk workgroup has left (the rows-through-LDS variants, round 4, need 4-8 more than the 224 of round 3)
    assert find("10k_panel256P")["VGPRs"] <= 232 and find("10k_panel256P")["Scratch"] == 0
    assert find("11k_panel256g")["VGPRs"] <= 224 and find("11k_panel256g")["Scratch"] == 0
    for variant in ("k_panel256gwILi4ELb0E", "k_panel256gwILi4ELb1E"):                                  # four waves on one CU: beside two bulk workgroups
        assert find(variant)["VGPRs"] <= 248 and find(variant)["Scratch"] == 0, variant
    assert find("7k_panelP")["VGPRs"] <= 224
    assert find("12k_potrf_diagP")["VGPRs"] <= 224
    assert find("16k_potrf_diag256g")["VGPRs"] <= 256
    chain = find("7k_chain")
    assert chain["VGPRs"] <= 256 and chain["Scratch"] <= 64, chain        # one wave per SIMD, a CU of its own: no spills to memory


def _build_rccl_host(exe):
    """gcc line of tests/c_host/shard_host_rccl.c (C99, HIP + RCCL public headers, the library's header); returns the CompletedProcess."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    rocm = "/opt/rocm"
    if not gcc or not os.path.exists(os.path.join(rocm, "include", "rccl", "rccl.h")):
        return None
    return subprocess.run([gcc, "-std=c99", "-pthread", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"),
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_host", "shard_host_rccl.c"), "-o", exe,
                           "-L" + os.path.join(rocm, "lib"), "-lrccl", "-lamdhip64", "-L" + os.path.join(ROOT, "gsum_amd"), "-lgsum_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "gsum_amd"), "-Wl,-rpath," + os.path.join(rocm, "lib")],
                          capture_output=True, text=True)


def test_rccl_c_host_compiles_and_links():
    """tests/c_host/shard_host_rccl.c -- INTEGRATION.md's multi-GPU recipe with a real communicator (ncclCommInitAll over the visible
    GPUs, the three in-place ncclAllGather calls verbatim) -- is valid C99 against the public headers and links against librccl,
    libamdhip64 and libgsum_hip.  It RUNS on the GPU box: tests/test_gpu_round4.py::test_rccl_c_host_gathers_the_sharded_scan."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        res = _build_rccl_host(os.path.join(tmp, "shard_host_rccl"))
        if res is None:
            pytest.skip("gcc or the RCCL headers are not installed")
        assert res.returncode == 0, res.stderr
    src = open(os.path.join(ROOT, "tests", "c_host", "shard_host_rccl.c")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for call in ("ncclAllGather(d_sld + lo, d_sld, c, ncclDouble, comm, stream)", "ncclAllGather(d_G + lo * k * k, d_G, c * k * k, ncclDouble, comm, stream)",
                 "ncclAllGather(d_inf + lo, d_inf, c, ncclInt64, comm, stream)"):
        assert call in doc                                                     # the documented recipe ...
        assert call.replace("d_sld", "d_sld[rank]").replace("d_G", "d_G[rank]").replace("d_inf", "d_inf[rank]").replace("lo", "lo[rank]") \
                   .replace("comm", "comm[rank]").replace("stream", "stream[rank]") in src      # ... is what the host runs, per rank


def test_c_host_of_the_sharded_scan_compiles_as_c99():
    """include/gsum_hip.h is a C header (no C++ in the boundary) and the C host of INTEGRATION.md's multi-GPU recipe,
    tests/c_host/shard_host.c, compiles against it with -std=c99 -Wall -Wextra -Werror and links against the library (every symbol
    it uses is exported).  It RUNS on the GPU box: tests/test_gpu_round3.py::test_c_host_sharded_scan_equals_unsharded."""
    import shutil
    import subprocess
    import tempfile
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("gcc not found")
    src = os.path.join(ROOT, "tests", "c_host", "shard_host.c")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "shard_host")
        res = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), src, "-o", exe,
                              "-L" + os.path.join(ROOT, "gsum_amd"), "-lgsum_hip", "-Wl,-rpath," + os.path.join(ROOT, "gsum_amd")],
                             capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        assert os.path.exists(exe)


def test_describe_thetas_equals_describing_the_clones():
    """The grid's descriptors are built without scikit-learn's per-theta clone (kernels.describe_thetas); they must be the
    clone's descriptors byte for byte -- values formed like Kernel.theta's setter forms them, theta laid out leaves left to right."""
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel as C, Matern, WhiteKernel
    from gsum_amd.kernels import describe_kernel, describe_thetas
    rng = np.random.RandomState(0)
    cases = [(C(1.0, "fixed") * RBF(0.2), 1), (RBF(0.2), 1), (C(2.0) * Matern([0.7, 1.3], nu=2.5) + WhiteKernel(1e-6, "fixed"), 2),
             (C(2.0) * Matern([0.7, 1.3], nu=1.5) + WhiteKernel(1e-6) + C(0.3), 2), (RBF([0.1, 0.2, 0.3]) * C(3.0) * C(0.5, "fixed"), 3),
             (C(2.0) * C(3.0) + Matern(0.4, nu=0.5), 1), (WhiteKernel(0.1) + C(1.5) * RBF(0.3, "fixed"), 1)]
    for kern, d in cases:
        thetas = kern.theta + rng.randn(32, len(kern.theta))
        got = describe_thetas(kern, thetas, d)
        want = [describe_kernel(kern.clone_with_theta(t), d) for t in thetas]
        assert all(bytes(a) == bytes(b) for a, b in zip(got, want)), kern
    with pytest.raises(ValueError, match="correct number of entries"):
        describe_thetas(RBF(0.2), [[0.1, 0.2]], 1)
    from sklearn.gaussian_process.kernels import PairwiseKernel
    assert describe_thetas(RBF(0.1) * RBF(0.2), [[0.1, 0.2]], 1)[0].is_tree              # (a tree since round 4)
    with pytest.raises(NotImplementedError, match="not supported on the device"):       # the family check comes before the theta-size check
        describe_thetas(C(1.0) * (PairwiseKernel() + WhiteKernel(0.1)), [[0.1, 0.2, 0.3, 0.4, 0.5]], 1)


def test_describe_gradients_equals_describing_the_clones():
    """Gradient parameters (code, dim, weight) per theta without the clone: equal to describe_gradient of the clone byte for byte,
    including the value-carrying weights of white / additive parameters and a kernel with nothing free."""
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel as C, Matern, WhiteKernel
    from gsum_amd.kernels import describe_gradient, describe_gradients
    rng = np.random.RandomState(1)
    cases = [(C(1.0, "fixed") * RBF(0.2), 1), (C(2.0) * Matern([0.7, 1.3], nu=2.5) + WhiteKernel(1e-6, "fixed"), 2),
             (C(2.0) * Matern([0.7, 1.3], nu=1.5) + WhiteKernel(1e-6) + C(0.3), 2), (RBF([0.1, 0.2, 0.3]) * C(3.0) * C(0.5, "fixed"), 3),
             (C(2.0) * C(3.0) + Matern(0.4, nu=0.5), 1), (WhiteKernel(0.1) + C(1.5) * RBF(0.3, "fixed"), 1),
             (C(1.0, "fixed") * RBF(0.2, "fixed"), 1)]
    for kern, d in cases:
        thetas = kern.theta + rng.randn(16, len(kern.theta))
        got = describe_gradients(kern, thetas, d)
        want = [describe_gradient(kern.clone_with_theta(t), d) for t in thetas]
        for a, b in zip(got, want):
            assert len(a) == len(b) == len(kern.theta)
            assert all(bytes(x) == bytes(y) for x, y in zip(a, b)), kern


def test_product_library_reads_no_environment_variable():
    """"Runtime configuration: none".  The product library's schedules are set through gsum_set_option alone; the GSUM_*
    overrides of rounds 1-3 (look-ahead, chain schedule, pivot guard) are compiled into the lab build only, and nothing reads
    GPU_MAX_HW_QUEUES."""
    import subprocess
    lib = os.path.join(ROOT, "gsum_amd", "libgsum_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    text = subprocess.run(["strings", "-n", "6", lib], capture_output=True, text=True).stdout
    for name in ("GSUM_LOOKAHEAD", "GSUM_CHAIN_PERSIST", "GSUM_PIVOT_GUARD_ULPS", "GPU_MAX_HW_QUEUES"):
        assert name not in text, name


def test_kernel_walk_cache_never_serves_an_edited_kernel():
    """kernels._compiled keeps the walk over a scikit-learn kernel per fingerprint (class of every node, parameters and bounds of every leaf):
    an objective evaluation of fit describes the same kernel at another theta tens of times (models.py:634-640).  A kernel edited in place is
    another key; an entry whose kernel was edited afterwards is not used for an equal, fresh kernel."""
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C
    from gsum_amd import kernels as K
    from gsum_amd.kernels import describe_gradient

    def make(flat):
        return (C(1.0) * RBF(0.5) + WhiteKernel(1e-6, noise_level_bounds="fixed")) if flat else \
            (C(1.0) * RBF(0.5) + C(0.5) * RBF([2.0, 3.0]) + WhiteKernel(1e-6, noise_level_bounds="fixed"))

    for flat in (True, False):
        K._COMPILED.clear()
        k = make(flat)
        d0 = bytes(gsum_amd.describe_kernel(k, 2))
        assert len(K._COMPILED) == 1 and bytes(gsum_amd.describe_kernel(k, 2)) == d0 and len(K._COMPILED) == 1
        th = k.theta + 0.2
        assert bytes(gsum_amd.describe_thetas(k, [th], 2)[0]) == bytes(gsum_amd.describe_kernel(k.clone_with_theta(th), 2))
        node = k
        while not isinstance(node, C):
            node = node.k1
        node.constant_value = 2.5                                         # edited in place: the cached walk's object now holds 2.5
        d1 = gsum_amd.describe_kernel(k, 2)
        assert (d1.amplitude if flat else d1.cval[0]) == 2.5
        assert bytes(gsum_amd.describe_kernel(make(flat), 2)) == d0       # a fresh kernel equal to the ORIGINAL: not the edited values
        node.constant_value_bounds = "fixed"                              # ... and a changed bound changes theta's layout
        assert len(describe_gradient(k, 2)) == len(k.theta) == (1 if flat else 4)
