"""Pin the CPU oracle (oracle/gsum_oracle.py) to outputs of the reference itself.

The golden files were produced by tests/golden/make_golden.py, which imports
buqeye/gsum read-only in the build container.  CPU-only; no GPU needed.
"""
import os

import numpy as np
import pytest

from conftest import load_golden, make_kernel, prior_kwargs, s5_inputs
from oracle import gsum_oracle as orc

RTOL = 1e-12


def test_helpers_against_reference(small_cases):
    h = small_cases["helpers"]
    y, ratio, ref = np.array(h["y"]), np.array(h["ratio"]), np.array(h["ref"])
    orders = np.array(h["orders"])
    np.testing.assert_array_equal(orc.coefficients(y, ratio, ref, orders), np.array(h["coefficients_arr"]))
    np.testing.assert_array_equal(orc.coefficients(y, 0.4, 2.0), np.array(h["coefficients_scalar"]))
    np.testing.assert_array_equal(orc.partials(y, ratio, ref, orders), np.array(h["partials_arr"]))
    np.testing.assert_array_equal(orc.partials(y, 0.4, 2.0), np.array(h["partials_scalar"]))
    x = np.array(h["geo_x"])
    np.testing.assert_array_equal(orc.geometric_sum(x, 0, np.inf), np.array(h["geo_0_inf"]))
    np.testing.assert_array_equal(orc.geometric_sum(x, 2, 5), np.array(h["geo_2_5"]))
    np.testing.assert_array_equal(orc.geometric_sum(x, 1, np.inf, excluded=[2, 7]), np.array(h["geo_1_inf_excl"]))
    np.testing.assert_array_equal(orc.geometric_sum(x, 3, 6, excluded=4), np.array(h["geo_3_6_excl"]))
    with pytest.raises(ValueError):
        orc.geometric_sum(x, 3, 2)
    with pytest.raises(ValueError):
        orc.coefficients(y[:, 0], 0.5)


def test_cgp_lml_fit_predict_against_reference(small_cases):
    assert len(small_cases["cgp"]) == 40
    for case in small_cases["cgp"]:
        kern = make_kernel(case["kernel"])
        pk = prior_kwargs(case["prior"])
        X, y = np.array(case["X"]), np.array(case["y"])
        theta = np.array(case["theta"])
        assert orc.cgp_lml(kern, theta, X, y, **pk) == pytest.approx(case["lml_theta"], rel=RTOL)
        assert orc.cgp_lml(kern, theta, X, y[:, 0], **pk) == pytest.approx(case["lml_1col"], rel=RTOL)
        fit = orc.cgp_fit(kern, X, y, **pk)
        g = case["fit"]
        assert fit["lml"] == pytest.approx(g["lml"], rel=RTOL)
        np.testing.assert_allclose(fit["center"], g["center"], rtol=RTOL, atol=1e-15)
        np.testing.assert_allclose(fit["disp"], g["disp"], rtol=RTOL, atol=1e-15)
        assert fit["df"] == g["df"]
        assert fit["scale"] == pytest.approx(g["scale"], rel=RTOL)
        assert fit["cov_factor"] == pytest.approx(g["cov_factor"], rel=RTOL)
        np.testing.assert_array_equal(fit["corr"][0], g["corr_row0"])
        np.testing.assert_allclose(fit["corr_L"][-1], g["corr_L_last"], rtol=1e-13, atol=1e-16)
        Xs, Xc, yc = np.array(case["Xs"]), np.array(case["Xc"]), np.array(case["yc"])
        p = case["predict"]
        m, s = orc.cgp_predict(fit, Xs, return_std=True)
        np.testing.assert_allclose(m, p["mean"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(s, p["std"], rtol=1e-9, atol=1e-12)
        _, cv = orc.cgp_predict(fit, Xs, return_cov=True)
        np.testing.assert_allclose(cv, p["cov"], rtol=1e-9, atol=1e-12)
        _, cvn = orc.cgp_predict(fit, Xs, return_cov=True, pred_noise=True)
        np.testing.assert_allclose(cvn, p["cov_noise"], rtol=1e-9, atol=1e-12)
        mc, sc = orc.cgp_predict(fit, Xs, return_std=True, Xc=Xc, y=yc)
        np.testing.assert_allclose(mc, p["mean_c"], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(sc, p["std_c"], rtol=1e-9, atol=1e-12)
        with pytest.raises(RuntimeError):
            orc.cgp_predict(fit, Xs, return_std=True, return_cov=True)


def test_trunc_lml_against_reference(small_cases):
    assert len(small_cases["trunc"]) == 36
    for case in small_cases["trunc"]:
        kern = make_kernel(case["kernel"])
        pk = prior_kwargs(case["prior"])
        X, y = np.array(case["X"]), np.array(case["y"])
        orders = np.array(case["orders"])
        for q, want in zip(case["ratios"], case["lml"]):
            got = orc.trunc_lml(kern, np.array(case["theta"]), X, y, orders, ratio=q, ref=case["ref"],
                                excluded=case["excluded"], **pk)
            assert got == pytest.approx(want, rel=RTOL)
        mask = ~np.isin(orders, case["excluded"])
        c = orc.coefficients(y, 0.5, case["ref"], orders)[:, mask]
        np.testing.assert_array_equal(c[0], case["coeffs_row0"])
        fit = orc.cgp_fit(kern, X, c, **pk)
        assert fit["cov_factor"] == pytest.approx(case["fit_cov_factor"], rel=RTOL)
        assert fit["lml"] == pytest.approx(case["fit_lml"], rel=RTOL)


def test_trunc_lml_array_ratio_ref(small_cases):
    a = small_cases["trunc_arrays"]
    kern = make_kernel(a["kernel"])
    X, y = np.array(a["X"]), np.array(a["y"])
    for s, want in zip(a["scales"], a["lml"]):
        got = orc.trunc_lml(kern, np.array(a["theta"]), X, y, np.array(a["orders"]),
                            ratio=s * np.array(a["ratio_arr"]), ref=np.array(a["ref_arr"]))
        assert got == pytest.approx(want, rel=RTOL)


def test_nonpd_behaviour(small_cases):
    c = small_cases["nonpd"]
    from sklearn.gaussian_process.kernels import RBF
    X, y = np.array(c["X"]), np.array(c["y"])
    assert c["lml_is_neg_inf"] and c["fit_raises_linalgerror"]
    assert orc.cgp_lml(RBF(1.0), np.log([1.0]), X, y, nugget=0) == -np.inf
    with pytest.raises(np.linalg.LinAlgError):
        orc.cgp_fit(RBF(1.0), X, y, nugget=0)


def test_notebook_grid_known_answer(notebook_grid):
    """The reference's one published likelihood known answer: MAP indices (36, 39)."""
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel
    g = notebook_grid
    assert g["argmax"] == [36, 39]
    assert g["best_Q"] == g["published"]["best_Q"] and g["best_ls"] == g["published"]["best_ls"]
    kern = RBF(0.2) + WhiteKernel(g["nugget"], noise_level_bounds="fixed")
    X, y = np.array(g["X_train"]), np.array(g["y_train"])
    grid = orc.lml_grid(kern, np.log(g["ls_vals"]), g["ratio_vals"], X, y, np.array(g["orders"]), ref=g["ref"])
    want = np.array(g["grid"])
    np.testing.assert_allclose(grid, want, rtol=1e-11)
    assert list(np.unravel_index(np.argmax(grid), grid.shape)) == [36, 39]
    assert not np.isneginf(grid).any()


def test_large_known_answers(large_lml):
    """S2-style input (dx = 0.5 ell) at n = 512 and 2048 (n = 8192 is checked on the GPU box)."""
    from sklearn.gaussian_process.kernels import RBF
    for case in large_lml:
        n, r = case["n"], case["r"]
        if n > 2048:
            continue
        X = case["dx"] * np.arange(n)[:, None]
        c = np.random.RandomState(case["seed"]).randn(n, r)
        y = orc.partials(c, ratio=0.5, ref=1.0, orders=np.arange(r))
        for q, want in case["lml"].items():
            got = orc.trunc_lml(RBF(case["length_scale"]), np.log([case["length_scale"]]), X, y, np.arange(r),
                                ratio=float(q), ref=1.0, nugget=case["nugget"])
            assert got == pytest.approx(want, rel=1e-12)


def test_truncation_mean_cov_predict_against_reference():
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C
    from conftest import load_golden
    g = load_golden("trunc_predict.json")
    kern = C(1.3) * RBF(0.35) + WhiteKernel(1e-6, noise_level_bounds="fixed")
    for case in g["cases"]:
        X, y, Xs = np.array(case["X"]), np.array(case["y"]), np.array(case["Xs"])
        orders, excluded, order = np.array(case["orders"]), case["excluded"], case["order"]
        pri = dict(center=0.1, disp=0, df=4, scale=1.2)
        # unfitted: prior centre and prior covariance factor
        f0 = orc.cov_factor(1.2 ** 2, 4)
        m0 = orc.trunc_predict_trunc(0.1, f0, kern, Xs, 2, 0.4, 3.0, excluded, fitted=False)
        np.testing.assert_allclose(m0, case["prior_mean"], rtol=1e-13)
        _, s0 = orc.trunc_predict_trunc(0.1, f0, kern, Xs, 2, 0.4, 3.0, excluded, fitted=False, return_std=True)
        np.testing.assert_allclose(s0, case["prior_std"], rtol=1e-13)
        mask = ~np.isin(orders, excluded)
        c = orc.coefficients(y, 0.4, 3.0, orders)[:, mask]
        fit = orc.cgp_fit(kern, X, c, **pri)
        m, sd = orc.trunc_predict_trunc(fit["center"], fit["cov_factor"], kern, Xs, order, 0.4, 3.0, excluded,
                                        return_std=True)
        np.testing.assert_allclose(m, case["mean"], rtol=1e-12)
        np.testing.assert_allclose(sd, case["std"], rtol=1e-11)
        _, cv = orc.trunc_predict_trunc(fit["center"], fit["cov_factor"], kern, Xs, order, 0.4, 3.0, excluded,
                                        return_cov=True)
        np.testing.assert_allclose(cv, case["cov"], rtol=1e-11)
        np.testing.assert_allclose(orc.trunc_mean(fit["center"], Xs, 0.4, 3.0, excluded=excluded), case["mean_0_inf"], rtol=1e-12)
        np.testing.assert_allclose(orc.trunc_cov(fit["cov_factor"], kern, Xs, Xs[:3], 0.4, 3.0, 2, 4, excluded),
                                   case["cov_2_4"], rtol=1e-11)
        np.testing.assert_allclose(orc.trunc_basis(Xs, 0.4, 3.0, start=1, excluded=excluded), case["basis_1_inf"], rtol=1e-13)
    p = g["cgp_prior"]
    Xs = np.array(p["Xs"])
    m, s = orc.cgp_prior_predict(RBF(0.5), Xs, center=0.2, df=5, scale=1.5, return_std=True)
    np.testing.assert_allclose(m, p["mean"])
    np.testing.assert_allclose(s, p["std"], rtol=1e-13)
    _, cv = orc.cgp_prior_predict(RBF(0.5), Xs, center=0.2, df=5, scale=1.5, return_cov=True)
    np.testing.assert_allclose(cv, p["cov"], rtol=1e-13)


def test_truncation_predict_all_kinds_against_reference():
    """TruncationGP.predict kind = interp / both / trunc, Xc / y overrides, array ratio / ref and constrained
    truncation (models.py:1389-1483): the oracle against the reference's outputs."""
    from conftest import load_golden, interp_case_setup
    g = load_golden("trunc_predict_interp.json")
    for case in g["cases"]:
        kern, ratio, ref = interp_case_setup(case)
        X, y, Xs = np.array(case["X"]), np.array(case["y"]), np.array(case["Xs"])
        orders, excluded, order = np.array(case["orders"]), case["excluded"], case["order"]
        rv = ratio(X) if callable(ratio) else ratio
        fv = ref(X) if callable(ref) else ref
        mask = ~np.isin(orders, excluded)
        c = orc.coefficients(y, rv, fv, orders)[:, mask]
        fit = orc.cgp_fit(kern, X, c, center=0.1, disp=0, df=4, scale=1.2)
        dX = None if case["dX"] is None else np.array(case["dX"])
        dy = None if case["dy"] is None else np.array(case["dy"])
        yo = np.squeeze(y[:, orders == order])
        tol = 1e-15 * case["cond_K_oo"] * 100 + 1e-12
        for kind, want in case["kinds"].items():
            kw = dict(excluded=excluded, kind=kind, dX=dX, dy=dy)
            m, sd = orc.trunc_predict(fit["center"], fit["cov_factor"], kern, Xs, order, ratio, ref, X, yo, return_std=True, **kw)
            np.testing.assert_allclose(m, want["mean"], rtol=tol, atol=tol)
            np.testing.assert_allclose(sd ** 2, np.array(want["std"]) ** 2, rtol=1e-9, atol=tol * np.max(np.array(want["std"]) ** 2))
            _, cv = orc.trunc_predict(fit["center"], fit["cov_factor"], kern, Xs, order, ratio, ref, X, yo, return_cov=True, **kw)
            np.testing.assert_allclose(cv, want["cov"], rtol=1e-9, atol=tol * np.abs(want["cov"]).max())
            np.testing.assert_allclose(orc.trunc_predict(fit["center"], fit["cov_factor"], kern, Xs, order, ratio, ref, X, yo, **kw),
                                       want["mean_only"], rtol=tol, atol=tol)
        sub = slice(None, None, case["subset"]["step"])
        m, sd = orc.trunc_predict(fit["center"], fit["cov_factor"], kern, Xs, order, ratio, ref, X[sub], yo[sub],
                                  excluded=excluded, kind="both", dX=dX, dy=dy, return_std=True)
        np.testing.assert_allclose(m, case["subset"]["mean"], rtol=tol, atol=tol)
        np.testing.assert_allclose(sd, case["subset"]["std"], rtol=1e-9)


def test_student_process_against_reference():
    """ConjugateStudentProcess / TruncationTP (models.py:1091-1273, 1519-1570): oracle vs reference outputs."""
    from conftest import load_golden, student_kernel
    g = load_golden("student.json")
    for case in g["cases"]:
        kern = student_kernel(case["kernel"])
        pri = case["priors"]
        X, c, Xs = np.array(case["X"]), np.array(case["c"]), np.array(case["Xs"])
        for th, want in zip(case["thetas"], case["lml"]):
            assert orc.csp_lml(kern, np.array(th), X, c, **pri) == pytest.approx(want, rel=1e-10)
        assert orc.csp_lml(kern, np.array(case["thetas"][0]), X, c[:, 0], **pri) == pytest.approx(case["lml_1d"], rel=1e-10)
        np.testing.assert_allclose(orc.csp_cov(kern, Xs, None, pri["df"], pri["scale"], pri["disp"]), case["unfit_cov"], rtol=1e-13)
        np.testing.assert_allclose(orc.csp_cov(kern, Xs, Xs[:2], pri["df"], pri["scale"], pri["disp"]), case["unfit_cov_cross"], rtol=1e-13)
        np.testing.assert_allclose(orc.csp_prior_predict(kern, Xs, return_std=True, **pri)[1], case["unfit_std"], rtol=1e-13)
        np.testing.assert_allclose(orc.csp_prior_predict(kern, Xs, return_cov=True, **pri)[1], case["unfit_cov_pred"], rtol=1e-13)
        fit = orc.cgp_fit(kern, X, c, **pri)
        f = case["fit"]
        np.testing.assert_allclose(fit["center"], f["center"], rtol=1e-9)
        np.testing.assert_allclose(fit["disp"], f["disp"], rtol=1e-10)
        assert fit["df"] == f["df"]
        assert fit["scale"] == pytest.approx(f["scale"], rel=1e-9)
        assert fit["cov_factor"] == pytest.approx(f["cov_factor"], rel=1e-9)
        assert orc.csp_lml(kern, None, X, c, **pri) == pytest.approx(f["lml_value"], rel=1e-10)
        p = case["predict"]
        m, sd = orc.csp_predict(fit, Xs, return_std=True)
        np.testing.assert_allclose(m, p["mean"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(sd, p["std"], rtol=1e-6)
        np.testing.assert_allclose(orc.csp_predict(fit, Xs, return_cov=True)[1], p["cov"], rtol=1e-6, atol=1e-9 * np.abs(p["cov"]).max())
        np.testing.assert_allclose(orc.csp_predict(fit, Xs), p["mean_only"], rtol=1e-8, atol=1e-10)
        ps = case["predict_subset"]
        sub = slice(None, None, ps["step"])
        m, sd = orc.csp_predict(fit, Xs, return_std=True, Xc=X[sub], y=c[sub])
        np.testing.assert_allclose(m, ps["mean"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(sd, ps["std"], rtol=1e-6)
        np.testing.assert_allclose(orc.csp_cov(kern, Xs, None, fit["df"], fit["scale"], fit["disp"]), case["cov"], rtol=1e-9)
        np.testing.assert_allclose(orc.csp_cov(kern, Xs, Xs[:2], fit["df"], fit["scale"], fit["disp"]), case["cov_cross"], rtol=1e-9)
        t = case["trunc"]
        y, orders = np.array(t["y"]), np.array(t["orders"])
        cc = orc.coefficients(y, t["ratio"], t["ref"], orders)
        det = np.sum(len(orders) * np.log(abs(t["ref"])) + np.sum(orders) * np.log(0.45) * np.ones(len(X)))
        c45 = orc.coefficients(y, 0.45, t["ref"], orders)
        assert orc.csp_lml(kern, np.array(case["thetas"][1]), X, c45, **pri) - det == pytest.approx(t["lml"], rel=1e-10)
        tfit = orc.cgp_fit(kern, X, cc, **pri)
        dX = None if t["dX"] is None else np.array(t["dX"])
        dy = None if t["dy"] is None else np.array(t["dy"])
        yo = y[:, orders == t["order"]][:, 0]
        tol = 1e-15 * t["cond_K_oo"] * 100 + 1e-12
        for kind, want in t["kinds"].items():
            kw = dict(kind=kind, dX=dX, dy=dy)
            m, sd = orc.ttp_predict(tfit, Xs, t["order"], t["ratio"], t["ref"], X, yo, return_std=True, **kw)
            np.testing.assert_allclose(m, want["mean"], rtol=tol, atol=tol)
            np.testing.assert_allclose(sd, want["std"], rtol=1e-6)
            _, cv = orc.ttp_predict(tfit, Xs, t["order"], t["ratio"], t["ref"], X, yo, return_cov=True, **kw)
            np.testing.assert_allclose(cv, want["cov"], rtol=1e-7, atol=tol * np.abs(want["cov"]).max())
            np.testing.assert_allclose(orc.ttp_predict(tfit, Xs, t["order"], t["ratio"], t["ref"], X, yo, **kw), want["mean_only"],
                                       rtol=tol, atol=tol)


def test_lml_gradient_against_reference():
    """log_marginal_likelihood(theta, eval_gradient=True): oracle vs the reference's values and gradients
    (ConjugateGaussianProcess), and vs finite differences of the reference's value path (ConjugateStudentProcess,
    whose own gradient path raises: see make_golden.gen_gradient)."""
    from conftest import load_golden, grad_kernel
    g = load_golden("gradient.json")
    for case in g["cases"]:
        kern = grad_kernel(case["kernel"])
        X, y, pri = np.array(case["X"]), np.array(case["y"]), case["priors"]
        for ev in case["evals"]:
            theta = np.array(ev["theta"])
            val, grad = orc.cgp_lml_grad(kern, theta, X, y, **pri)
            assert val == pytest.approx(ev["lml"], rel=1e-10)
            np.testing.assert_allclose(grad, ev["grad"], rtol=1e-8, atol=1e-9 * np.abs(ev["grad"]).max())
            if "student_grad_fd" in ev:
                val, grad = orc.csp_lml_grad(kern, theta, X, y, **pri)
                assert val == pytest.approx(ev["student_lml"], rel=1e-10)
                # the fixture is a finite difference of a value with rounding noise ~ eps cond(R)
                tol = 3e-14 * np.linalg.cond(kern(X)) + 1e-8
                np.testing.assert_allclose(grad, ev["student_grad_fd"], rtol=tol, atol=tol * np.abs(ev["student_grad_fd"]).max())


def _config4_inputs(g):
    from sklearn.gaussian_process.kernels import RBF
    n, r = g["n"], g["r"]
    X = g["dx"] * np.arange(n)[:, None]
    K = RBF(g["length_scale"])(X)
    K[np.diag_indices_from(K)] += g["nugget"]
    c = np.linalg.cholesky(K) @ np.random.RandomState(g["seed"]).randn(n, r)
    return X, orc.partials(c, ratio=0.5, ref=1.0, orders=np.arange(r))


def test_cbar_ratio_grid_against_reference():
    """BASELINE config 4 in miniature (tests/golden/cbar_ratio_grid.json, written by the reference): the per-point
    restatement, the one-factorisation checker used at n = 8192, argmax interior and equal."""
    from conftest import load_golden
    from sklearn.gaussian_process.kernels import RBF
    g = load_golden("cbar_ratio_grid.json")
    X, y = _config4_inputs(g)
    want = np.array(g["grid_ratio_by_cbar"])
    theta = np.log([g["length_scale"]])
    orders = np.arange(g["r"])
    full = orc.cbar_ratio_grid(RBF(0.2), theta, X, y, orders, g["ratios"], g["cbars"])
    one = orc.cbar_ratio_grid_one_factor(RBF(0.2), theta, X, y, orders, g["ratios"], g["cbars"])
    np.testing.assert_allclose(full, want, rtol=1e-11)
    np.testing.assert_allclose(one, want, rtol=1e-11)
    am = list(np.unravel_index(np.argmax(one), one.shape))
    assert am == g["argmax"] and 0 < am[0] < len(g["ratios"]) - 1 and 0 < am[1] < len(g["cbars"]) - 1
    strip = np.array([[orc.trunc_lml(RBF(0.2), np.log([e]), X, y, orders, ratio=q) for e in g["ells"]] for q in g["ratios"]])
    np.testing.assert_allclose(strip, np.array(g["strip_ratio_by_ell"]), rtol=1e-10)
    assert list(np.unravel_index(np.argmax(strip), strip.shape)) == g["strip_argmax"]


def test_oracle_s1_plumbing_golden():
    """BASELINE config 1 (X = linspace(0, 1, 128), 4 orders, RBF(0.2), coefficients drawn by the recipe of
    datasets.py:65-71): the oracle against the reference's own TruncationGP / ConjugateGaussianProcess outputs
    (tests/golden/s1_plumbing.json, written by make_golden.py from /root/reference).  cond(K) = 5.6e11 here, so this
    also says the restatement issues the same LAPACK calls on the same matrix: the values agree to the last bit."""
    from sklearn.gaussian_process.kernels import RBF
    d = load_golden("s1_plumbing.json")
    X = np.linspace(0, 1, d["n"])[:, None]
    y, orders = np.array(d["y"]), np.arange(d["r"])
    for e in d["lml"]:
        got = orc.trunc_lml(RBF(e["ell"]), np.log([e["ell"]]), X, y, orders, ratio=e["ratio"], ref=d["ref"])
        assert got == pytest.approx(e["value"], rel=1e-12)
    c = orc.coefficients(y, d["ratio"], d["ref"], orders)
    fit = orc.cgp_fit(RBF(d["length_scale"]), X, c)
    assert fit["cov_factor"] == pytest.approx(d["cgp"]["cov_factor"], rel=1e-12)
    mean, std = orc.cgp_predict(fit, np.array(d["cgp"]["Xs"]), return_std=True)
    np.testing.assert_allclose(mean, d["cgp"]["mean"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(std, d["cgp"]["std"], rtol=1e-9)          # sqrt of a difference of O(1) numbers at 1e-11
    assert orc.cgp_lml(RBF(d["length_scale"]), np.log([d["length_scale"]]), X, c) == pytest.approx(d["cgp"]["lml"], rel=1e-12)


@pytest.mark.skipif(os.environ.get("GSUM_RUN_SLOW") != "1",
                    reason="n = 16384 on the CPU: ~4 minutes and ~12 GB; run with GSUM_RUN_SLOW=1 (log of the last run: "
                           "profiles/r03_oracle_s5_check.log)")
def test_oracle_s5_predict_golden_slow():
    """BASELINE config 5 exactly as specified (S5: n = 16384 2-D points, Matern-5/2(ell = [0.7, 1.3]) + White(1e-6), 8 curves):
    the oracle's fit / log-likelihood / predictive mean and standard deviation at 16 probe points against the reference's
    (tests/golden/s5_predict.json; models.py:671-738, 912-1039, 753-845).  Pins the oracle at the largest BASELINE size."""
    from sklearn.gaussian_process.kernels import Matern, WhiteKernel
    d = load_golden("s5_predict.json")
    X, Xp, y = s5_inputs(d)
    kern = Matern(length_scale=d["length_scale"], nu=2.5) + WhiteKernel(d["white"], noise_level_bounds="fixed")
    fit = orc.cgp_fit(kern, X, y)
    assert fit["cov_factor"] == pytest.approx(d["cov_factor"], rel=1e-12)
    mean, std = orc.cgp_predict(fit, Xp, return_std=True)
    np.testing.assert_allclose(mean, d["mean"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(std ** 2, np.array(d["std"]) ** 2, rtol=1e-10, atol=1e-10 * d["cov_factor"])
    lml = orc.cgp_lml(kern, np.log(d["length_scale"]), X, y)
    assert lml == pytest.approx(d["lml"], rel=1e-12)
    print("oracle vs reference at S5: lml rel", abs(lml - d["lml"]) / abs(d["lml"]), "mean max abs", np.abs(mean - d["mean"]).max(),
          "var max abs / cov_factor", np.abs(std ** 2 - np.array(d["std"]) ** 2).max() / d["cov_factor"])
