#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the reference (buqeye/gsum) itself.

Runs ONLY in the build container, where /root/reference is mounted read-only.
The fixtures it writes are data (seeded inputs + the reference's outputs); the
reference never travels.  Usage:  python tests/golden/make_golden.py

``import gsum`` needs three non-numeric packages that are absent here
(docrep: docstring templating; seaborn: plotting; statsmodels' MVT class used
only by gsum.diagnostics).  As SURVEY.md §8(c) records, they are replaced by
in-memory placeholder modules; no numeric code is stubbed.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("GSUM_REFERENCE", "/root/reference")


def _import_reference():
    d = types.ModuleType("docrep")

    class _DP:
        def __init__(self, *a, **k):
            pass

        def get_sectionsf(self, *a, **k):
            return lambda f: f

        def dedent(self, f):
            return f

    d.DocstringProcessor = _DP
    sys.modules["docrep"] = d
    sys.modules["seaborn"] = types.ModuleType("seaborn")
    for name in ("statsmodels", "statsmodels.sandbox", "statsmodels.sandbox.distributions",
                 "statsmodels.sandbox.distributions.mv_normal"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["statsmodels.sandbox.distributions.mv_normal"].MVT = object
    sys.path.insert(0, REF)
    import gsum  # noqa
    return gsum


gsum = _import_reference()
from sklearn.gaussian_process.kernels import RBF, Matern, WhiteKernel, ConstantKernel as C  # noqa: E402


def L(a):
    return np.asarray(a, dtype=float).tolist()


# ---------------------------------------------------------------------------
# kernel zoo: spec -> sklearn kernel (tests rebuild kernels from the spec)
# ---------------------------------------------------------------------------

def make_kernel(spec):
    fam = spec["family"]
    ls = spec["length_scale"]
    ls = ls if np.ndim(ls) == 0 else np.asarray(ls, dtype=float)
    if fam == "rbf":
        k = RBF(length_scale=ls)
    elif fam == "matern52":
        k = Matern(length_scale=ls, nu=2.5)
    elif fam == "matern32":
        k = Matern(length_scale=ls, nu=1.5)
    else:
        raise ValueError(fam)
    if spec.get("amplitude") is not None:
        k = C(spec["amplitude"]) * k
    if spec.get("white") is not None:
        k = k + WhiteKernel(spec["white"], noise_level_bounds="fixed")
    if spec.get("additive") is not None:
        k = k + C(spec["additive"], constant_value_bounds="fixed")
    return k


KERNELS_1D = [
    dict(name="rbf", family="rbf", length_scale=0.2),
    dict(name="matern52", family="matern52", length_scale=0.3),
    dict(name="c_rbf_white", family="rbf", length_scale=0.25, amplitude=1.7, white=1e-3),
    dict(name="rbf_white_nb", family="rbf", length_scale=0.2, white=1e-10),
    dict(name="matern32_add", family="matern32", length_scale=0.4, additive=0.5),
]
KERNELS_2D = [
    dict(name="rbf_aniso", family="rbf", length_scale=[0.7, 1.3]),
    dict(name="matern52_aniso_white", family="matern52", length_scale=[0.7, 1.3], white=1e-6),
    dict(name="rbf_iso_2d", family="rbf", length_scale=0.9),
]
PRIORS = [
    dict(name="default", center=0, disp=0, df=1, scale=1),
    dict(name="center_df", center=0.3, disp=0, df=3, scale=1.5),
    dict(name="disp", center=0.2, disp=2.0, df=1, scale=1),
    dict(name="disp_sd", center=-0.1, disp=0.7, sd=1.2),
    dict(name="sd_only", center=-0.4, disp=0, sd=0.8),
]


def prior_kwargs(p):
    return {k: v for k, v in p.items() if k != "name"}


def gen_cgp_cases():
    """ConjugateGaussianProcess: lml, fit attributes, predict (models.py:671-1039)."""
    cases = []
    for dim, kernels in ((1, KERNELS_1D), (2, KERNELS_2D)):
        rng = np.random.RandomState(100 + dim)
        n, r = 24, 3
        if dim == 1:
            X = np.sort(rng.rand(n))[:, None] * 2.0
            Xs = np.linspace(-0.1, 2.1, 7)[:, None]
        else:
            X = rng.rand(n, 2) * np.array([3.0, 5.0])
            Xs = rng.rand(6, 2) * np.array([3.0, 5.0])
        y = rng.randn(n, r) + 0.3
        Xc = X[::3] + 0.01
        yc = rng.randn(len(Xc), r)
        for ks in kernels:
            for pr in PRIORS:
                kern = make_kernel(ks)
                gp = gsum.ConjugateGaussianProcess(kernel=kern, optimizer=None, **prior_kwargs(pr))
                theta = kern.theta + 0.1          # evaluate away from the spec's own theta
                lml_theta = gp.log_marginal_likelihood(theta=theta, X=X, y=y)
                lml_1col = gp.log_marginal_likelihood(theta=theta, X=X, y=y[:, 0])
                gp.fit(X, y)
                mean_s, std_s = gp.predict(Xs, return_std=True)
                _, cov_s = gp.predict(Xs, return_cov=True)
                _, cov_noise = gp.predict(Xs, return_cov=True, pred_noise=True)
                mean_c, std_c = gp.predict(Xs, return_std=True, Xc=Xc, y=yc)
                cases.append(dict(
                    kernel=ks, prior=pr, X=L(X), y=L(y), Xs=L(Xs), Xc=L(Xc), yc=L(yc),
                    theta=L(theta), lml_theta=float(lml_theta), lml_1col=float(lml_1col),
                    fit=dict(lml=float(gp.log_marginal_likelihood_value_), center=L(gp.center_),
                             disp=L(gp.disp_), df=float(gp.df_), scale=float(gp.scale_),
                             cov_factor=float(gp.cov_factor_),
                             corr_row0=L(gp.corr_[0]), corr_L_last=L(gp.corr_L_[-1]),
                             prior_cov_probe=L(gp.cov(Xs[:3], Xs[3:5]))),
                    predict=dict(mean=L(mean_s), std=L(std_s), cov=L(cov_s), cov_noise=L(cov_noise),
                                 mean_c=L(mean_c), std_c=L(std_c)),
                ))
    return cases


def gen_trunc_cases():
    """TruncationGP.log_marginal_likelihood / fit (models.py:1367-1387, 1485-1507)."""
    cases = []
    rng = np.random.RandomState(7)
    n = 30
    X = np.linspace(0, 3, n)[:, None]
    for orders, excluded in (([0, 1, 2, 3], None), ([0, 2, 3, 4, 5], None), ([0, 1, 2, 3, 4], [1])):
        orders = np.array(orders)
        c = rng.randn(n, len(orders))
        for ref in (1.0, 10.0):
            y = gsum.partials(c, ratio=0.5, ref=ref, orders=orders)
            for pr in (PRIORS[0], PRIORS[2], PRIORS[4]):
                for ks in (KERNELS_1D[0], KERNELS_1D[2]):
                    kern = make_kernel(ks)
                    gp = gsum.TruncationGP(kernel=kern, ratio=0.5, ref=ref, excluded=excluded,
                                           optimizer=None, **prior_kwargs(pr))
                    gp.fit(X, y, orders=orders)
                    ratios = [0.3, 0.45, 0.5, 0.62]
                    theta = kern.theta - 0.2
                    lmls = [float(gp.log_marginal_likelihood(theta=theta, ratio=q)) for q in ratios]
                    cases.append(dict(kernel=ks, prior=pr, X=L(X), y=L(y), orders=orders.tolist(),
                                      excluded=excluded, ref=ref, ratios=ratios, theta=L(theta), lml=lmls,
                                      coeffs_row0=L(gp.coeffs_[0]),
                                      fit_cov_factor=float(gp.coeffs_process.cov_factor_),
                                      fit_lml=float(gp.coeffs_process.log_marginal_likelihood_value_)))
    # per-point ratio / ref arrays through callables (models.py:1309-1317)
    orders = np.arange(4)
    c = rng.randn(n, 4)
    ratio_arr = 0.3 + 0.2 * X[:, 0] / 3
    ref_arr = 5.0 + X[:, 0]
    y = gsum.partials(c, ratio=ratio_arr, ref=ref_arr, orders=orders)
    kern = make_kernel(KERNELS_1D[0])
    gp = gsum.TruncationGP(kernel=kern, ratio=lambda X_, scale=1.0: scale * (0.3 + 0.2 * X_[:, 0] / 3),
                           ref=lambda X_: 5.0 + X_[:, 0], optimizer=None)
    gp.fit(X, y, orders=orders)
    scales = [0.8, 1.0, 1.1]
    lmls = [float(gp.log_marginal_likelihood(theta=kern.theta, scale=s)) for s in scales]
    arr_case = dict(X=L(X), y=L(y), orders=orders.tolist(), ratio_arr=L(ratio_arr), ref_arr=L(ref_arr),
                    kernel=KERNELS_1D[0], theta=L(kern.theta), scales=scales, lml=lmls)
    return cases, arr_case


def gen_helpers():
    """helpers.py:71-182 outputs on seeded inputs."""
    rng = np.random.RandomState(11)
    y = rng.randn(5, 4)
    ratio = rng.rand(5) * 0.5 + 0.2
    ref = rng.rand(5) + 1.0
    orders = np.array([0, 2, 3, 5])
    out = dict(y=L(y), ratio=L(ratio), ref=L(ref), orders=orders.tolist())
    out["coefficients_arr"] = L(gsum.coefficients(y, ratio, ref, orders))
    out["coefficients_scalar"] = L(gsum.coefficients(y, 0.4, 2.0))
    out["partials_arr"] = L(gsum.partials(y, ratio, ref, orders))
    out["partials_scalar"] = L(gsum.partials(y, 0.4, 2.0))
    x = rng.rand(3, 2) * 0.8
    out["geo_x"] = L(x)
    out["geo_0_inf"] = L(gsum.geometric_sum(x, 0, np.inf))
    out["geo_2_5"] = L(gsum.geometric_sum(x, 2, 5))
    out["geo_1_inf_excl"] = L(gsum.geometric_sum(x, 1, np.inf, excluded=[2, 7]))
    out["geo_3_6_excl"] = L(gsum.geometric_sum(x, 3, 6, excluded=4))
    return out


def gen_notebook_grid():
    """The published 80x100 (Q, ell) scan; known answer: argmax (36, 39).

    docs/notebooks/correlated_EFT_publication.ipynb:134-171, 978-1005, 1029,
    1265-1266, 1444-1459, 1611-1612.
    """
    x = np.linspace(0, 1, 100)
    X = x[:, None]
    orders = np.arange(0, 4)
    ls, sd, center, ref, ratio, nugget, seed = 0.2, 1, 0, 10, 0.5, 1e-10, 3
    kernel = RBF(length_scale=ls, length_scale_bounds="fixed") + \
        WhiteKernel(noise_level=nugget, noise_level_bounds="fixed")
    gp = gsum.ConjugateGaussianProcess(kernel=kernel, center=center, df=np.inf, scale=sd, nugget=0)
    coeffs_all = -gp.sample_y(X, n_samples=21, random_state=seed)
    data_all = gsum.partials(coeffs_all, ratio, ref=ref, orders=np.arange(21))
    data = data_all[:, :4]
    mask = np.array([(i - 1) % 24 == 0 for i in range(len(x))])
    kernel_fit = RBF(length_scale=ls) + WhiteKernel(noise_level=nugget, noise_level_bounds="fixed")
    gp_trunc = gsum.TruncationGP(kernel=kernel_fit, ref=ref, ratio=ratio, center=0, disp=0, df=1, scale=1,
                                 optimizer=None)
    gp_trunc.fit(X[mask], y=data[mask], orders=orders)
    ls_vals = np.linspace(1e-3, 0.5, 100)
    ratio_vals = np.linspace(0.3, 0.7, 80)
    grid = np.array([[gp_trunc.log_marginal_likelihood(theta=[ls_, ], ratio=q) for ls_ in np.log(ls_vals)]
                     for q in ratio_vals])
    like = np.exp(grid - np.max(grid))
    i, j = np.unravel_index(np.argmax(like), like.shape)
    return dict(X_train=L(X[mask]), y_train=L(data[mask]), orders=orders.tolist(), ref=ref,
                ls_vals=L(ls_vals), ratio_vals=L(ratio_vals), nugget=nugget,
                argmax=[int(i), int(j)], best_Q=float(ratio_vals[i]), best_ls=float(ls_vals[j]),
                n_neg_inf=int(np.isneginf(grid).sum()), grid=L(grid),
                published=dict(best_Q=0.4822784810126582, best_ls=0.19757575757575757))


def gen_large():
    """S2/S3 known answers: X = 0.1*arange(n), RBF(0.2), nugget 1e-10."""
    out = []
    for n, r in ((512, 4), (2048, 4), (8192, 6)):
        X = 0.1 * np.arange(n)[:, None]
        c = np.random.RandomState(0).randn(n, r)
        y = gsum.partials(c, ratio=0.5, ref=1.0, orders=np.arange(r))
        gp = gsum.TruncationGP(kernel=RBF(0.2), ratio=0.5, ref=1.0, center=0, disp=0, df=1, scale=1,
                               optimizer=None)
        gp.X_train_, gp.y_train_, gp.orders_ = X, y, np.arange(r)
        vals = {}
        for q in (0.5, 0.45):
            vals[str(q)] = float(gp.log_marginal_likelihood(theta=np.log([0.2]), ratio=q))
        out.append(dict(n=n, r=r, dx=0.1, length_scale=0.2, nugget=1e-10, seed=0, lml=vals))
        print("large", n, r, vals, flush=True)
    return out


def gen_large_gp_drawn():
    """Same X / kernel as gen_large, but coefficients drawn FROM the GP (c = L z), the statistically
    faithful variant of SURVEY.md §8d.  The quadratic form is then O(n), not O(n / lambda_min), and the
    log-likelihood is well conditioned: this is the input class the 1e-10 parity bar is checked on."""
    out = []
    for n, r in ((512, 4), (2048, 4), (8192, 6)):
        X = 0.1 * np.arange(n)[:, None]
        K = RBF(0.2)(X)
        K[np.diag_indices_from(K)] += 1e-10
        c = np.linalg.cholesky(K) @ np.random.RandomState(1).randn(n, r)
        y = gsum.partials(c, ratio=0.5, ref=1.0, orders=np.arange(r))
        gp = gsum.TruncationGP(kernel=RBF(0.2), ratio=0.5, ref=1.0, center=0, disp=0, df=1, scale=1,
                               optimizer=None)
        gp.X_train_, gp.y_train_, gp.orders_ = X, y, np.arange(r)
        vals = {}
        for q in (0.5, 0.45):
            vals[str(q)] = float(gp.log_marginal_likelihood(theta=np.log([0.2]), ratio=q))
        out.append(dict(n=n, r=r, dx=0.1, length_scale=0.2, nugget=1e-10, seed=1, lml=vals,
                        y_checksum=float(np.sum(y * np.cos(np.arange(y.size).reshape(y.shape)))),
                        recipe="c = cholesky(RBF(0.2)(X) + 1e-10 I) @ RandomState(1).randn(n, r)"))
        print("gp-drawn", n, r, vals, flush=True)
    return out


def gen_trunc_predict():
    """TruncationGP.predict(kind='trunc'), mean/cov/basis scaling and the unfitted (prior) paths
    (models.py:1337-1365, 1389-1483) plus ConjugateGaussianProcess prior predict (:792-793)."""
    rng = np.random.RandomState(21)
    n = 20
    X = np.linspace(0, 2, n)[:, None]
    Xs = np.linspace(0.05, 1.95, 6)[:, None]
    out = []
    for orders, excluded in (([0, 1, 2, 3], None), ([0, 2, 3, 4, 5], [3])):
        orders = np.array(orders)
        c = rng.randn(n, len(orders))
        y = gsum.partials(c, ratio=0.4, ref=3.0, orders=orders)
        kern = C(1.3) * RBF(0.35) + WhiteKernel(1e-6, noise_level_bounds="fixed")
        gp = gsum.TruncationGP(kernel=kern, ratio=0.4, ref=3.0, excluded=excluded, center=0.1, disp=0, df=4, scale=1.2,
                               optimizer=None)
        prior_mean = gp.predict(Xs, order=2)                      # unfitted: underlying_properties
        _, prior_std = gp.predict(Xs, order=2, return_std=True)
        gp.fit(X, y, orders=orders)
        order = int(orders[2])
        m, sd = gp.predict(Xs, order=order, return_std=True, kind="trunc")
        _, cv = gp.predict(Xs, order=order, return_cov=True, kind="trunc")
        out.append(dict(X=L(X), y=L(y), Xs=L(Xs), orders=orders.tolist(), excluded=excluded, order=order,
                        prior_mean=L(prior_mean), prior_std=L(prior_std), mean=L(m), std=L(sd), cov=L(cv),
                        mean_0_inf=L(gp.mean(Xs)), cov_2_4=L(gp.cov(Xs, Xs[:3], start=2, end=4)),
                        basis_1_inf=L(gp.basis(Xs, start=1))))
    cgp = gsum.ConjugateGaussianProcess(kernel=RBF(0.5), center=0.2, df=5, scale=1.5, optimizer=None)
    pm, ps = cgp.predict(Xs, return_std=True)
    _, pc = cgp.predict(Xs, return_cov=True)
    return dict(cases=out, cgp_prior=dict(Xs=L(Xs), mean=L(pm), std=L(ps), cov=L(pc)))


def gen_trunc_predict_interp():
    """TruncationGP.predict with kind='interp' / 'both', Xc / y overrides, position-dependent ratio and ref, and a
    truncation error constrained by (dX, dy) (models.py:1389-1483).  Conditioning sets are kept well separated:
    the reference conditions on the un-jittered cov(Xc, Xc) (quirk Q7) and cond(K_oo) is recorded per case."""
    rng = np.random.RandomState(33)
    Xs = np.linspace(0.03, 1.97, 7)[:, None]
    specs = [
        dict(name="const_ratio_rbf", n=8, orders=[0, 1, 2, 3], excluded=None, order_idx=2,
             kernel=dict(const=1.3, base="rbf", ls=0.35, white=1e-6), ratio=0.4, ref=3.0, constrained=False),
        dict(name="excluded_rbf", n=7, orders=[0, 2, 3, 4, 5], excluded=[3], order_idx=3,
             kernel=dict(const=0.8, base="rbf", ls=0.3, white=1e-8), ratio=0.55, ref=1.5, constrained=False),
        dict(name="array_ratio_matern", n=12, orders=[0, 1, 2, 3, 4], excluded=None, order_idx=2,
             kernel=dict(const=1.0, base="matern25", ls=0.5, white=None), ratio="array", ref="array", constrained=False),
        dict(name="constrained_trunc", n=8, orders=[0, 1, 2, 3], excluded=None, order_idx=1,
             kernel=dict(const=1.1, base="rbf", ls=0.4, white=1e-8), ratio=0.45, ref=2.0, constrained=True),
    ]
    out = []
    for sp in specs:
        n = sp["n"]
        X = np.linspace(0, 2, n)[:, None]
        orders = np.array(sp["orders"])
        kd = sp["kernel"]
        base = RBF(kd["ls"]) if kd["base"] == "rbf" else Matern(kd["ls"], nu=2.5)
        kern = C(kd["const"]) * base
        if kd["white"] is not None:
            kern = kern + WhiteKernel(kd["white"], noise_level_bounds="fixed")
        if sp["ratio"] == "array":
            ratio = lambda X: 0.3 + 0.1 * X[:, 0]          # noqa: E731
            ref = lambda X: 2.0 + X[:, 0]                  # noqa: E731
            ratio_v, ref_v = ratio(X), ref(X)
        else:
            ratio, ref = sp["ratio"], sp["ref"]
            ratio_v, ref_v = ratio, ref
        c = rng.randn(n, len(orders))
        y = gsum.partials(c, ratio=ratio_v, ref=ref_v, orders=orders)
        gp = gsum.TruncationGP(kernel=kern, ratio=ratio, ref=ref, excluded=sp["excluded"], center=0.1, disp=0, df=4,
                               scale=1.2, optimizer=None)
        dX = dy = None
        if sp["constrained"]:
            dX, dy = np.array([[0.5], [1.5]]), np.array([0.0, 0.02])
        gp.fit(X, y, orders=orders, dX=dX, dy=dy)
        order = int(orders[sp["order_idx"]])
        case = dict(name=sp["name"], X=L(X), y=L(y), Xs=L(Xs), orders=orders.tolist(), excluded=sp["excluded"],
                    order=order, kernel=kd, ratio=sp["ratio"], ref=sp["ref"], constrained=sp["constrained"],
                    dX=None if dX is None else L(dX), dy=None if dy is None else L(dy), kinds={})
        K_oo = gp.cov(start=0, end=order, X=X, Xp=X)
        case["cond_K_oo"] = float(np.linalg.cond(K_oo))
        for kind in ("interp", "both", "trunc"):
            m, sd = gp.predict(Xs, order=order, return_std=True, kind=kind)
            _, cv = gp.predict(Xs, order=order, return_cov=True, kind=kind)
            m_only = gp.predict(Xs, order=order, kind=kind)
            case["kinds"][kind] = dict(mean=L(m), std=L(sd), cov=L(cv), mean_only=L(m_only))
        # conditioning on a subset with an explicit y (models.py:1401-1406)
        sub = slice(None, None, 2)
        yo = np.squeeze(y[:, orders == order])[sub]
        m, sd = gp.predict(Xs, order=order, return_std=True, Xc=X[sub], y=yo, kind="both")
        case["subset"] = dict(step=2, mean=L(m), std=L(sd))
        out.append(case)
        print("trunc-predict-interp", sp["name"], "cond", case["cond_K_oo"], flush=True)
    return dict(cases=out)


STUDENT_SPECS = [
    dict(name="rbf_white_disp0", kernel=dict(const=None, base="rbf", ls=0.5, white=1e-6), n=9,
         priors=dict(center=0.3, disp=0, df=4, scale=1.1), constrained=False),
    dict(name="rbf_white_disp2", kernel=dict(const=None, base="rbf", ls=0.5, white=1e-6), n=9,
         priors=dict(center=0.3, disp=2.0, df=4, scale=1.1), constrained=True),
    dict(name="matern_disp05", kernel=dict(const=1.5, base="matern25", ls=0.7, white=None), n=12,
         priors=dict(center=-0.2, disp=0.5, df=3.5, scale=0.8), constrained=False),
]


def student_kernel(kd):
    base = RBF(kd["ls"]) if kd["base"] == "rbf" else Matern(kd["ls"], nu=2.5)
    kern = base if kd["const"] is None else C(kd["const"]) * base
    if kd["white"] is not None:
        kern = kern + WhiteKernel(kd["white"], noise_level_bounds="fixed")
    return kern


def gen_student():
    """ConjugateStudentProcess (models.py:1091-1273) and TruncationTP (:1519-1570): likelihood values, fitted
    hyperparameters, cov, predict (fitted / unfitted / Xc-y override) and the truncation predict kinds."""
    rng = np.random.RandomState(77)
    Xs = np.linspace(0.07, 1.93, 6)[:, None]
    out = []
    for sp in STUDENT_SPECS:
        n = sp["n"]
        X = np.linspace(0, 2, n)[:, None]
        kern = student_kernel(sp["kernel"])
        pri = sp["priors"]
        Lc = np.linalg.cholesky(kern(X) + 1e-8 * np.eye(n))
        c = 0.4 + Lc @ rng.randn(n, 3)                     # smooth coefficient curves
        case = dict(name=sp["name"], kernel=sp["kernel"], priors=pri, X=L(X), c=L(c), Xs=L(Xs))
        gp = gsum.ConjugateStudentProcess(kernel=kern, optimizer=None, **pri)
        case["unfit_cov"] = L(gp.cov(Xs))
        case["unfit_cov_cross"] = L(gp.cov(Xs, Xs[:2]))
        case["unfit_std"] = L(gp.predict(Xs, return_std=True)[1])
        case["unfit_cov_pred"] = L(gp.predict(Xs, return_cov=True)[1])
        thetas = [kern.theta, kern.theta + 0.3]
        case["thetas"] = [L(t) for t in thetas]
        case["lml"] = [float(gp.log_marginal_likelihood(theta=t, X=X, y=c)) for t in thetas]
        case["lml_1d"] = float(gp.log_marginal_likelihood(theta=thetas[0], X=X, y=c[:, 0]))
        gp.fit(X, c)
        case["fit"] = dict(center=L(gp.center_), disp=L(gp.disp_), df=float(gp.df_), scale=float(gp.scale_),
                           cov_factor=float(gp.cov_factor_), lml_value=float(gp.log_marginal_likelihood_value_))
        m, sd = gp.predict(Xs, return_std=True)
        _, cv = gp.predict(Xs, return_cov=True)
        case["predict"] = dict(mean=L(m), std=L(sd), cov=L(cv), mean_only=L(gp.predict(Xs)))
        m, sd = gp.predict(Xs, return_std=True, Xc=X[::2], y=c[::2])
        case["predict_subset"] = dict(step=2, mean=L(m), std=L(sd))
        case["cov"] = L(gp.cov(Xs))
        case["cov_cross"] = L(gp.cov(Xs, Xs[:2]))
        # truncation layer
        orders = np.arange(3)
        ypart = gsum.partials(c, ratio=0.5, ref=2.0, orders=orders)
        tp = gsum.TruncationTP(kernel=kern, ratio=0.5, ref=2.0, optimizer=None, **pri)
        dX = dy = None
        if sp["constrained"]:
            dX, dy = np.array([[0.4], [1.6]]), np.array([0.01, -0.02])
        tp.fit(X, ypart, orders=orders, dX=dX, dy=dy)
        tcase = dict(y=L(ypart), orders=orders.tolist(), ratio=0.5, ref=2.0, dX=None if dX is None else L(dX),
                     dy=None if dy is None else L(dy), order=1,
                     lml=float(tp.log_marginal_likelihood(theta=thetas[1], ratio=0.45)), kinds={})
        for kind in ("both", "interp", "trunc"):
            m, sd = tp.predict(Xs, order=1, return_std=True, kind=kind)
            _, cv = tp.predict(Xs, order=1, return_cov=True, kind=kind)
            tcase["kinds"][kind] = dict(mean=L(m), std=L(sd), cov=L(cv), mean_only=L(tp.predict(Xs, order=1, kind=kind)))
        tcase["cond_K_oo"] = float(np.linalg.cond(tp.cov(X=X, Xp=X, start=0, end=1)))
        case["trunc"] = tcase
        out.append(case)
        print("student", sp["name"], case["lml"], "cond K_oo", tcase["cond_K_oo"], flush=True)
    return dict(cases=out)


GRAD_SPECS = [
    dict(name="rbf_1d_white_fixed", d=1, n=24, kernel=dict(const=None, base="rbf", ls=0.3, white=1e-6, white_fixed=True, add=None),
         priors=dict(center=0, disp=0, df=1, scale=1)),
    dict(name="const_rbf_white_fixed", d=1, n=24, kernel=dict(const=1.4, base="rbf", ls=0.25, white=1e-8, white_fixed=True, add=None),
         priors=dict(center=0.3, disp=2.0, df=3, scale=1.2)),
    dict(name="const_matern52_aniso_white_add", d=2, n=30,
         kernel=dict(const=1.3, base="matern25", ls=[0.5, 0.7], white=0.01, white_fixed=False, add=0.2),
         priors=dict(center=0.2, disp=1.5, df=3, scale=1.2)),
    dict(name="matern32_iso_2d", d=2, n=30, kernel=dict(const=None, base="matern15", ls=0.6, white=1e-6, white_fixed=True, add=None),
         priors=dict(center=-0.1, disp=0, df=2.5, scale=0.9)),
    dict(name="matern12_aniso", d=2, n=20, kernel=dict(const=0.8, base="matern05", ls=[0.4, 0.9], white=None, white_fixed=True, add=None),
         priors=dict(center=0, disp=0.7, df=4, scale=1.0)),
    dict(name="rbf_aniso_sd_prior", d=3, n=30, kernel=dict(const=None, base="rbf", ls=[0.5, 0.8, 1.1], white=1e-6, white_fixed=True, add=None),
         priors=dict(center=0.1, disp=0.5, sd=1.3)),
]


def grad_kernel(kd):
    nu = {"rbf": None, "matern25": 2.5, "matern15": 1.5, "matern05": 0.5}[kd["base"]]
    ls = kd["ls"] if np.ndim(kd["ls"]) == 0 else np.array(kd["ls"], dtype=float)
    base = RBF(ls) if nu is None else Matern(ls, nu=nu)
    kern = base if kd["const"] is None else C(kd["const"]) * base
    if kd["white"] is not None:
        kern = kern + (WhiteKernel(kd["white"], noise_level_bounds="fixed") if kd["white_fixed"] else WhiteKernel(kd["white"]))
    if kd["add"] is not None:
        kern = kern + C(kd["add"])
    return kern


def gen_gradient():
    """log_marginal_likelihood(theta, eval_gradient=True): value and gradient from the reference's
    ConjugateGaussianProcess (models.py:957-1056).  The reference's ConjugateStudentProcess gradient path raises
    (kernel(X, eval_gradient) at :1204 passes True as Y), so for it the fixture holds a Richardson-extrapolated
    central difference of the reference's VALUE path, which pins the intended formulas of :1227-1272."""
    rng = np.random.RandomState(99)
    out = []
    for sp in GRAD_SPECS:
        n, d = sp["n"], sp["d"]
        X = np.sort(rng.rand(n, d), axis=0) * 2.0
        kern = grad_kernel(sp["kernel"])
        Lc = np.linalg.cholesky(kern(X) + 1e-8 * np.eye(n))
        y = 0.25 + Lc @ rng.randn(n, 3)
        pri = sp["priors"]
        gp = gsum.ConjugateGaussianProcess(kernel=kern, optimizer=None, **pri)
        sp_ = gsum.ConjugateStudentProcess(kernel=kern, optimizer=None, **pri)
        case = dict(name=sp["name"], kernel=sp["kernel"], priors=pri, X=L(X), y=L(y), evals=[])
        for shift in (0.0, 0.2, -0.15):
            theta = kern.theta + shift
            val, grad = gp.log_marginal_likelihood(theta, eval_gradient=True, X=X, y=y)
            ev = dict(theta=L(theta), lml=float(val), grad=L(grad))
            if "sd" not in pri:            # df0 = inf: the Student normalisation is inf - inf (nan in the reference too)
                f = lambda t: sp_.log_marginal_likelihood(t, X=X, y=y)   # noqa: E731
                fd = np.zeros_like(theta)
                for p in range(len(theta)):
                    def cd(h):
                        tp, tm = theta.copy(), theta.copy()
                        tp[p] += h
                        tm[p] -= h
                        return (f(tp) - f(tm)) / (2 * h)
                    h = 2e-4
                    fd[p] = (4 * cd(h / 2) - cd(h)) / 3
                ev["student_lml"] = float(f(theta))
                ev["student_grad_fd"] = L(fd)
            case["evals"].append(ev)
        out.append(case)
        print("gradient", sp["name"], case["evals"][0]["lml"], case["evals"][0]["grad"], flush=True)
    return dict(cases=out)


def gen_nonpd():
    """Cholesky failure -> -inf (models.py:968-972); fit raises (models.py:711)."""
    X = np.array([[0.0], [0.5], [0.5], [1.0]])
    y = np.array([[0.1], [0.2], [0.2], [0.3]])
    gp = gsum.ConjugateGaussianProcess(kernel=RBF(1.0), nugget=0, optimizer=None)
    v = gp.log_marginal_likelihood(theta=np.log([1.0]), X=X, y=y)
    raised = False
    try:
        gp.fit(X, y)
    except np.linalg.LinAlgError:
        raised = True
    return dict(X=L(X), y=L(y), lml_is_neg_inf=bool(np.isneginf(v)), fit_raises_linalgerror=raised)


def gen_classmethods():
    """The reference's public classmethod surface (models.py:170-503) on a host Cholesky factor: values and the
    eval_gradient=True outputs, for a one- and a two-column basis and both dispersion regimes."""
    rng = np.random.RandomState(77)
    n = 18
    X = np.sort(rng.rand(n))[:, None] * 4.0
    kern = C(1.3) * RBF(0.5)
    R, dR = kern(X, eval_gradient=True)
    R[np.diag_indices_from(R)] += 1e-6
    chol = np.linalg.cholesky(R)
    y = rng.randn(n, 3) + 0.2
    out = dict(X=L(X), y=L(y), kernel=dict(amplitude=1.3, length_scale=0.5), nugget=1e-6, cases=[])
    cgp = gsum.ConjugateGaussianProcess
    for basis, center0, disp0 in (
            (np.ones((n, 1)), np.array([0.3]), np.array([[2.0]])),
            (np.ones((n, 1)), np.array([-0.2]), np.array([[0.0]])),
            (np.concatenate([np.ones((n, 1)), X], axis=1), np.array([0.3, -0.2]), np.array([[2.0, 0.3], [0.3, 1.0]]))):
        for df0, scale0 in ((3.0, 1.5), (np.inf, 0.7)):
            c, dc = cgp.compute_center(y, chol, basis, center0, disp0, 'cholesky', eval_gradient=True, dR=dR)
            V, dV = cgp.compute_disp(y, chol, basis, disp0, 'cholesky', eval_gradient=True, dR=dR)
            df, ddf = cgp.compute_df(y, df0, eval_gradient=True, dR=dR)
            s2, ds2 = cgp.compute_scale_sq(y, chol, basis, center0, disp0, df0, scale0, 'cholesky',
                                           eval_gradient=True, dR=dR)
            out["cases"].append(dict(
                basis_cols=int(basis.shape[1]), center0=L(center0), disp0=L(disp0),
                df0=("inf" if np.isinf(df0) else df0), scale0=scale0,
                center=L(c), d_center=L(dc), disp=L(V), d_disp=L(dV), df=float(df), d_df=L(ddf),
                scale_sq=float(s2), d_scale_sq=L(ds2),
                cov_factor=float(cgp.compute_cov_factor(s2, df)),
                center_1d_y=L(cgp.compute_center(y[:, 0], chol, basis, center0, disp0, 'cholesky')),
                scale_sq_1d_y=float(cgp.compute_scale_sq(y[:, 0], chol, basis, center0, disp0, df0, scale0, 'cholesky'))))
    B = rng.randn(n, 4)
    w, Q = np.linalg.eigh(R)
    out["solve_sqrt"] = dict(B=L(B), chol=L(cgp.solve_sqrt(chol, B, 'cholesky')),
                             eig_tuple=L(cgp.solve_sqrt((w, Q), B, 'eig')),
                             eig_sqrt=L(cgp.solve_sqrt(Q * np.sqrt(w), B, 'eig')),
                             vec=L(cgp.solve_sqrt(chol, B[:, 0], 'cholesky')))
    out["num_y"] = [int(cgp.num_y(y)), int(cgp.num_y(y[:, 0]))]
    out["avg_y"] = L(cgp.avg_y(y))
    return out


def gen_cbar_ratio_grid():
    """BASELINE config 4 in miniature: the (cbar, ratio) likelihood surface, cbar realised the only way the reference
    allows (sd=cbar: df0 = inf, scale0 = cbar; models.py:115-117, 419-422), every entry one call of
    TruncationGP.log_marginal_likelihood.  Coefficients are drawn from the GP with cbar = 1, ratio = 0.5, so the
    maximum is interior.  Also an (ell, ratio) strip with the default prior."""
    n, r = 192, 5
    X = 0.1 * np.arange(n)[:, None]
    K = RBF(0.2)(X)
    K[np.diag_indices_from(K)] += 1e-10
    c = np.linalg.cholesky(K) @ np.random.RandomState(5).randn(n, r)
    y = gsum.partials(c, ratio=0.5, ref=1.0, orders=np.arange(r))
    cbars = np.geomspace(0.25, 4, 9)
    ratios = np.linspace(0.3, 0.7, 9)
    theta = np.log([0.2])
    grid = np.empty((len(ratios), len(cbars)))
    for b, cbar in enumerate(cbars):
        gp = gsum.TruncationGP(kernel=RBF(0.2), ratio=0.5, ref=1.0, center=0, disp=0, sd=cbar, optimizer=None)
        gp.X_train_, gp.y_train_, gp.orders_ = X, y, np.arange(r)
        for a, q in enumerate(ratios):
            grid[a, b] = gp.log_marginal_likelihood(theta=theta, ratio=q)
    ells = np.linspace(0.12, 0.3, 7)
    gp = gsum.TruncationGP(kernel=RBF(0.2), ratio=0.5, ref=1.0, center=0, disp=0, df=1, scale=1, optimizer=None)
    gp.X_train_, gp.y_train_, gp.orders_ = X, y, np.arange(r)
    strip = np.array([[gp.log_marginal_likelihood(theta=np.log([e]), ratio=q) for e in ells] for q in ratios])
    return dict(n=n, r=r, dx=0.1, length_scale=0.2, nugget=1e-10, seed=5,
                recipe="c = cholesky(RBF(0.2)(X) + 1e-10 I) @ RandomState(5).randn(n, r); y = partials(c, 0.5, 1.0)",
                cbars=L(cbars), ratios=L(ratios), grid_ratio_by_cbar=L(grid),
                argmax=[int(v) for v in np.unravel_index(np.argmax(grid), grid.shape)],
                ells=L(ells), strip_ratio_by_ell=L(strip),
                strip_argmax=[int(v) for v in np.unravel_index(np.argmax(strip), strip.shape)])


def gen_s5_predict():
    """BASELINE config 5 exactly as SURVEY.md 8(d) S5 states it: n = 16384 points in 2-D (the box bench.py uses: mean
    nearest-neighbour spacing ~0.5 ell), Matern-5/2(ell = [0.7, 1.3]) + White(1e-6) fixed, 8 curves.  The reference's
    fit -> log_marginal_likelihood -> predict(return_std) (models.py:671-738, 912-1039, 753-845) at the first 16 of the
    m = 2048 new points bench.py's predict leg uses, so that leg can check itself against this file."""
    n, r, m = 16384, 8, 2048
    side = np.array([0.35, 0.65]) * np.sqrt(n)
    X = np.random.RandomState(0).rand(n, 2) * side
    Xs = np.random.RandomState(1).rand(m, 2) * side
    y = np.random.RandomState(2).randn(n, r)
    kern = Matern(length_scale=[0.7, 1.3], nu=2.5) + WhiteKernel(1e-6, noise_level_bounds="fixed")
    gp = gsum.ConjugateGaussianProcess(kernel=kern, center=0, disp=0, df=1, scale=1, optimizer=None)
    import time
    t0 = time.time()
    gp.fit(X, y)
    print("s5 fit", time.time() - t0, flush=True)
    lml = float(gp.log_marginal_likelihood(theta=np.log([0.7, 1.3])))
    print("s5 lml", lml, time.time() - t0, flush=True)
    mean, std = gp.predict(Xs[:16], return_std=True)
    print("s5 predict", time.time() - t0, flush=True)
    mean_n, std_n = gp.predict(Xs[:16], return_std=True, pred_noise=True)
    # round 4: the configuration's full set of new points is 16384 = 8 shards of 2048 (one per GPU); RandomState(1).rand(16384, 2)
    # continues the stream whose first 2048 rows are Xs above.  32 probes in EVERY shard (every 64th point of the shard).
    m_all, shards, per = 16384, 8, 32
    Xs_all = np.random.RandomState(1).rand(m_all, 2) * side
    assert np.array_equal(Xs_all[:m], Xs)
    probe_idx = np.concatenate([g * (m_all // shards) + (m_all // shards // per) * np.arange(per) for g in range(shards)])
    mean_s, std_s = gp.predict(Xs_all[probe_idx], return_std=True)
    _, std_sn = gp.predict(Xs_all[probe_idx], return_std=True, pred_noise=True)
    print("s5 shard probes", time.time() - t0, flush=True)
    shard_probes = dict(m_all=m_all, shards=shards, per_shard=per, index=[int(i) for i in probe_idx], mean=L(mean_s), std=L(std_s),
                        std_pred_noise=L(std_sn),
                        recipe="Xs_all = RandomState(1).rand(16384, 2) * side; shard g = rows [2048 g, 2048 (g + 1)); probes = every 64th row of a shard")
    return dict(n=n, r=r, m=m, probes=16, shard_probes=shard_probes, side=L(side), length_scale=[0.7, 1.3], white=1e-6, nugget=1e-10,
                recipe="X = RandomState(0).rand(n,2)*side; Xs = RandomState(1).rand(m,2)*side; y = RandomState(2).randn(n,r); "
                       "side = [0.35, 0.65]*sqrt(n); ConjugateGaussianProcess(Matern([0.7,1.3],2.5)+White(1e-6,fixed), "
                       "center=0, disp=0, df=1, scale=1, optimizer=None)",
                lml=lml, cov_factor=float(gp.cov_factor_), scale=float(gp.scale_), df=float(gp.df_),
                center=L(gp.center_), mean=L(mean), std=L(std), std_pred_noise=L(std_n),
                y_checksum=float(np.sum(y * np.cos(np.arange(y.size).reshape(y.shape)))),
                X_checksum=float(np.sum(X * np.cos(np.arange(X.size).reshape(X.shape)))))


def gen_s1_plumbing():
    """BASELINE config 1: X = linspace(0, 1, 128), 4 orders, RBF(0.2), ratio 0.5, ref 1, coefficients by
    the recipe of datasets.py:65-71 (multivariate normal draw, nugget 1e-10, random_state 0); TruncationGP fit, lml at a few
    (ell, ratio) points, predict at 9 new points.  cond(K) ~ 1e13 here: tests scale their tolerance by it."""
    n, r = 128, 4
    X = np.linspace(0, 1, n)[:, None]
    kern = RBF(0.2)
    y = gsum.make_gaussian_partial_sums(X, orders=r, kernel=kern, ratio=0.5, ref=1.0, nugget=1e-10, random_state=0)
    gp = gsum.TruncationGP(kernel=RBF(0.2), ratio=0.5, ref=1.0, center=0, disp=0, df=1, scale=1, optimizer=None)
    gp.fit(X, y, orders=np.arange(r))
    K = kern(X)
    K[np.diag_indices_from(K)] += 1e-10
    out = dict(n=n, r=r, y=L(y), length_scale=0.2, ratio=0.5, ref=1.0, nugget=1e-10, cond=float(np.linalg.cond(K)),
               cov_factor=float(gp.coeffs_process.cov_factor_), scale=float(gp.coeffs_process.scale_),
               df=float(gp.coeffs_process.df_), lml=[])
    for ell in (0.2, 0.15, 0.3):
        for q in (0.5, 0.4):
            out["lml"].append(dict(ell=ell, ratio=q, value=float(gp.log_marginal_likelihood(theta=np.log([ell]), ratio=q))))
    cgp = gsum.ConjugateGaussianProcess(kernel=RBF(0.2), center=0, disp=0, df=1, scale=1, optimizer=None)
    c = gsum.coefficients(y, ratio=0.5, ref=1.0, orders=np.arange(r))
    cgp.fit(X, c)
    Xs = np.linspace(0.03, 0.97, 9)[:, None]
    mean, std = cgp.predict(Xs, return_std=True)
    out["cgp"] = dict(Xs=L(Xs), mean=L(mean), std=L(std), cov_factor=float(cgp.cov_factor_),
                      lml=float(cgp.log_marginal_likelihood(theta=np.log([0.2]))))
    return out


def gen_underlying():
    """BaseConjugateProcess.underlying_properties (models.py:740-749): mean(X) [, sqrt(diag(cov(X))) | cov(X)] of the process --
    prior quantities before fit, center_ / cov_factor_-scaled after.  Inputs stored with the outputs."""
    cases = []
    for dim, kernels in ((1, KERNELS_1D[:2]), (2, KERNELS_2D[:1])):
        rng = np.random.RandomState(300 + dim)
        n, r = 20, 3
        if dim == 1:
            X = np.sort(rng.rand(n))[:, None] * 2.0
            Xs = np.linspace(-0.1, 2.1, 5)[:, None]
        else:
            X = rng.rand(n, 2) * np.array([3.0, 5.0])
            Xs = rng.rand(5, 2) * np.array([3.0, 5.0])
        y = rng.randn(n, r) + 0.3
        for ks in kernels:
            for pr in PRIORS:
                gp = gsum.ConjugateGaussianProcess(kernel=make_kernel(ks), optimizer=None, **prior_kwargs(pr))
                prior = None
                try:                                   # before fit: prior mean / covariance (center0, scale0 ...), if the reference allows it
                    pm, pc = gp.underlying_properties(Xs, return_cov=True)
                    prior = dict(mean=L(pm), cov=L(pc))
                except Exception as exc:               # noqa: BLE001 -- recorded, not hidden
                    prior = dict(error=type(exc).__name__)
                gp.fit(X, y)
                m0 = gp.underlying_properties(Xs)
                m1, sd = gp.underlying_properties(Xs, return_std=True)
                m2, cv = gp.underlying_properties(Xs, return_cov=True)
                m3, cv_both = gp.underlying_properties(Xs, return_std=True, return_cov=True)     # return_cov wins (:742-744)
                cases.append(dict(kernel=ks, prior=pr, X=L(X), y=L(y), Xs=L(Xs), unfitted=prior,
                                  mean=L(m0), mean_std=L(m1), std=L(sd), mean_cov=L(m2), cov=L(cv),
                                  both_returns_cov=bool(np.shape(cv_both) == np.shape(cv) and np.array_equal(cv_both, cv)),
                                  cov_factor=float(gp.cov_factor_), center=L(gp.center_)))
    return cases


def gen_sample_y():
    """BaseConjugateProcess.sample_y (models.py:847-879), default path: rng.multivariate_normal on predict's / underlying_properties'
    mean and covariance, one call per curve.  One curve (1-D mean) and three curves (stacked (m, r, n_samples))."""
    cases = []
    rng = np.random.RandomState(41)
    n = 16
    X = np.sort(rng.rand(n))[:, None] * 2.0
    Xs = np.array([0.05, 0.37, 0.81, 1.33, 1.96, 2.4])[:, None]
    for r in (1, 3):
        y = np.sin(3.0 * X) * (1.0 + 0.2 * np.arange(r)) + 0.05 * rng.randn(n, r)
        y_in = y[:, 0] if r == 1 else y
        gp = gsum.ConjugateGaussianProcess(kernel=make_kernel(KERNELS_1D[0]), optimizer=None, center=0.1, disp=0, df=3, scale=1.5,
                                           nugget=1e-8)
        gp.fit(X, y_in)
        _, cov_p = gp.predict(Xs, return_cov=True)
        for underlying in (False, True):
            ys = gp.sample_y(Xs, n_samples=4, random_state=11, underlying=underlying)
            cases.append(dict(kernel=KERNELS_1D[0], X=L(X), y=L(y_in), Xs=L(Xs), r=r, underlying=underlying, n_samples=4, random_state=11,
                              samples=L(ys), shape=list(np.shape(ys)), cov_factor=float(gp.cov_factor_),
                              predict_cov_min_eig=float(np.linalg.eigvalsh(cov_p).min())))
    return cases


TREE_KERNELS = [          # (expression, input dimension): scikit-learn kernels OUTSIDE the flattened family (missing 2)
    ("RBF(0.6) + RBF(2.5)", 1),
    ("C(2.0) * RBF(0.8) + C(0.5) * Matern(1.5, nu=2.5) + WhiteKernel(1e-3)", 1),
    ("RationalQuadratic(length_scale=0.9, alpha=1.3)", 2),
    ("(RBF(0.7) + C(0.1)) * Matern([1.0, 2.0], nu=1.5) + WhiteKernel(1e-4, noise_level_bounds='fixed')", 2),
    ("RationalQuadratic(length_scale=1.1, alpha=0.7) * RBF([0.9, 1.7]) + C(0.3, constant_value_bounds='fixed')", 2),
    ("C(1.5) * Matern(0.8, nu=0.5) + RBF(1.9)", 1),
    # round 5: the remaining stationary scikit-learn kernels with a unit diagonal, and the Exponentiation operator
    ("C(1.2) * ExpSineSquared(length_scale=1.1, periodicity=3.0) * RBF(4.0) + WhiteKernel(1e-3)", 1),
    ("Exponentiation(RBF(0.8), 2.0) + C(0.2) + WhiteKernel(1e-4)", 1),
    ("Matern([1.2, 2.1], nu=np.inf) * C(1.5) + WhiteKernel(1e-3)", 2),
    ("Exponentiation(C(1.1) * RationalQuadratic(length_scale=1.2, alpha=0.7) + C(0.5), 2) + WhiteKernel(1e-3)", 2),
    ("ExpSineSquared(length_scale=0.9, periodicity=6.0) * C(0.8) + WhiteKernel(1e-2)", 1),
    # ... and the one scikit-learn leaf that is not stationary (its diagonal depends on the points)
    ("C(0.001) * DotProduct(sigma_0=2.0) ** 2 + RBF(1.1) + WhiteKernel(1e-3)", 1),
    ("C(0.05) * DotProduct(sigma_0=1.3) * RBF([2.5, 4.0]) + WhiteKernel(1e-2)", 2),
    # ... and a kernel without any leaf: c 1 1^T + w I
    ("C(0.5) + WhiteKernel(0.3)", 1),
]


def tree_kernel(expr):
    from sklearn.gaussian_process.kernels import DotProduct, Exponentiation, ExpSineSquared, RationalQuadratic
    return eval(expr, {"__builtins__": {}}, dict(RBF=RBF, Matern=Matern, WhiteKernel=WhiteKernel, C=C, RationalQuadratic=RationalQuadratic,
                                                 ExpSineSquared=ExpSineSquared, Exponentiation=Exponentiation, DotProduct=DotProduct, np=np))


def gen_tree_kernels():
    """General Sum / Product kernel trees through the reference (models.py:146-147, 686-688, 958-960 accept any scikit-learn kernel):
    log_marginal_likelihood value + gradient at two thetas, the fitted hyperparameters with the optimiser off, predict mean / std,
    one kernel matrix row, and a TruncationGP likelihood.  Points well separated (conditioned matrices): 1e-10 is meaningful."""
    cases = []
    for idx, (expr, dim) in enumerate(TREE_KERNELS):
        rng = np.random.RandomState(500 + idx)
        n, r = 60, 3
        X = (np.sort(rng.rand(n))[:, None] * 25.0) if dim == 1 else rng.rand(n, 2) * np.array([9.0, 14.0])
        Xs = (np.linspace(0.5, 24.0, 9)[:, None]) if dim == 1 else rng.rand(9, 2) * np.array([9.0, 14.0])
        y = rng.randn(n, r)
        kern = tree_kernel(expr)
        pri = dict(center=0.2, disp=0.5, df=3, scale=1.3)
        gp = gsum.ConjugateGaussianProcess(kernel=kern, optimizer=None, nugget=1e-8, **pri)
        evals = []
        for shift in (0.0, 0.15):
            theta = kern.theta + shift * np.cos(np.arange(len(kern.theta)))
            val, grad = gp.log_marginal_likelihood(theta, eval_gradient=True, X=X, y=y)
            evals.append(dict(theta=L(theta), lml=float(val), grad=L(grad)))
        gp.fit(X, y)
        mean, std = gp.predict(Xs, return_std=True)
        K = kern(X)
        tg = gsum.TruncationGP(kernel=kern, ratio=0.6, ref=2.0, optimizer=None, nugget=1e-8, **pri)
        orders = np.arange(r)
        yp = gsum.partials(y, ratio=0.6, ref=2.0, orders=orders)
        tg.fit(X, yp, orders=orders)
        cases.append(dict(expr=expr, dim=dim, X=L(X), y=L(y), Xs=L(Xs), priors=pri, nugget=1e-8, evals=evals, cond=float(np.linalg.cond(K + 1e-8 * np.eye(n))),
                          fit=dict(lml=float(gp.log_marginal_likelihood_value_), center=L(gp.center_), disp=L(gp.disp_), df=float(gp.df_),
                                   scale=float(gp.scale_), cov_factor=float(gp.cov_factor_)),
                          predict=dict(mean=L(mean), std=L(std)), K_row3=L(K[3]), K_cross_row3=L(kern(X, Xs)[3]),
                          trunc=dict(ratio=0.55, lml=float(tg.log_marginal_likelihood(theta=kern.theta, ratio=0.55)))))
    return cases


def gen_eig_mode():
    """decomposition='eig' through the reference (models.py:713-717, 810-811, 973-974, 1016-1019): likelihood value + gradient, fit,
    predict (mean, std, cov) and the square-root attributes' defining property, for a conjugate GP, a Student process and a truncation GP."""
    cases = []
    for idx, (expr, dim) in enumerate([("C(1.2) * RBF(0.8) + WhiteKernel(1e-3)", 1), ("Matern([0.9, 1.6], nu=2.5) + WhiteKernel(1e-2)", 2)]):
        rng = np.random.RandomState(900 + idx)
        n, r = 40, 3
        X = (np.sort(rng.rand(n))[:, None] * 8.0) if dim == 1 else rng.rand(n, 2) * np.array([5.0, 7.0])
        Xs = (np.linspace(0.3, 7.5, 7)[:, None]) if dim == 1 else rng.rand(7, 2) * np.array([5.0, 7.0])
        y = rng.randn(n, r)
        kern = tree_kernel(expr)
        pri = dict(center=0.2, disp=0.5, df=3, scale=1.3)
        gp = gsum.ConjugateGaussianProcess(kernel=kern, optimizer=None, nugget=1e-8, decomposition='eig', **pri)
        theta = kern.theta + 0.1 * np.cos(np.arange(len(kern.theta)))
        val, grad = gp.log_marginal_likelihood(theta, eval_gradient=True, X=X, y=y)
        gp.fit(X, y)
        mean, std = gp.predict(Xs, return_std=True)
        _, cov = gp.predict(Xs, return_cov=True)
        S = gp.corr_sqrt_
        sp = gsum.ConjugateStudentProcess(kernel=kern, optimizer=None, nugget=1e-8, decomposition='eig', **pri)
        sval = sp.log_marginal_likelihood(theta, X=X, y=y)
        tg = gsum.TruncationGP(kernel=kern, ratio=0.6, ref=2.0, optimizer=None, nugget=1e-8, decomposition='eig', **pri)
        orders = np.arange(r)
        tg.fit(X, gsum.partials(y, ratio=0.6, ref=2.0, orders=orders), orders=orders)
        cases.append(dict(expr=expr, dim=dim, X=L(X), y=L(y), Xs=L(Xs), priors=pri, nugget=1e-8, theta=L(theta), lml=float(val), grad=L(grad),
                          fit=dict(lml=float(gp.log_marginal_likelihood_value_), center=L(gp.center_), disp=L(gp.disp_), df=float(gp.df_),
                                   scale=float(gp.scale_), cov_factor=float(gp.cov_factor_)),
                          predict=dict(mean=L(mean), std=L(std), cov=L(cov)),
                          sqrt_residual=float(np.abs(S @ S.T - (gp.corr_ + 1e-8 * np.eye(n))).max()),
                          student_lml=float(sval), trunc_lml=float(tg.log_marginal_likelihood(theta=kern.theta, ratio=0.55))))
    return cases


def main():
    only = set(sys.argv[1:])           # e.g. `make_golden.py classmethods cbar_ratio_grid` regenerates just those files
    if only:
        gens = dict(classmethods=gen_classmethods, cbar_ratio_grid=gen_cbar_ratio_grid, s5_predict=gen_s5_predict,
                    s1_plumbing=gen_s1_plumbing, underlying=gen_underlying, sample_y=gen_sample_y, tree_kernels=gen_tree_kernels,
                    eig_mode=gen_eig_mode)
        for name in only:
            with open(os.path.join(HERE, name + ".json"), "w") as f:
                json.dump(gens[name](), f, indent=1)
        return
    with open(os.path.join(HERE, "classmethods.json"), "w") as f:
        json.dump(gen_classmethods(), f, indent=1)
    with open(os.path.join(HERE, "cbar_ratio_grid.json"), "w") as f:
        json.dump(gen_cbar_ratio_grid(), f, indent=1)
    out = {}
    out["helpers"] = gen_helpers()
    out["cgp"] = gen_cgp_cases()
    tc, arr = gen_trunc_cases()
    out["trunc"] = tc
    out["trunc_arrays"] = arr
    out["nonpd"] = gen_nonpd()
    with open(os.path.join(HERE, "small_cases.json"), "w") as f:
        json.dump(out, f)
    with open(os.path.join(HERE, "notebook_grid.json"), "w") as f:
        json.dump(gen_notebook_grid(), f)
    with open(os.path.join(HERE, "trunc_predict.json"), "w") as f:
        json.dump(gen_trunc_predict(), f)
    with open(os.path.join(HERE, "trunc_predict_interp.json"), "w") as f:
        json.dump(gen_trunc_predict_interp(), f)
    with open(os.path.join(HERE, "gradient.json"), "w") as f:
        json.dump(gen_gradient(), f)
    with open(os.path.join(HERE, "student.json"), "w") as f:
        json.dump(gen_student(), f)
    with open(os.path.join(HERE, "large_lml.json"), "w") as f:
        json.dump(gen_large(), f, indent=1)
    with open(os.path.join(HERE, "large_lml_gp_drawn.json"), "w") as f:
        json.dump(gen_large_gp_drawn(), f, indent=1)
    with open(os.path.join(HERE, "s1_plumbing.json"), "w") as f:
        json.dump(gen_s1_plumbing(), f, indent=1)
    with open(os.path.join(HERE, "underlying.json"), "w") as f:
        json.dump(gen_underlying(), f, indent=1)
    with open(os.path.join(HERE, "sample_y.json"), "w") as f:
        json.dump(gen_sample_y(), f, indent=1)
    with open(os.path.join(HERE, "tree_kernels.json"), "w") as f:
        json.dump(gen_tree_kernels(), f, indent=1)
    with open(os.path.join(HERE, "eig_mode.json"), "w") as f:
        json.dump(gen_eig_mode(), f, indent=1)
    with open(os.path.join(HERE, "s5_predict.json"), "w") as f:      # ~10 minutes on 8 cores, ~15 GB
        json.dump(gen_s5_predict(), f, indent=1)
    import sklearn, scipy
    with open(os.path.join(HERE, "VERSIONS.json"), "w") as f:
        json.dump(dict(numpy=np.__version__, scipy=scipy.__version__, sklearn=sklearn.__version__,
                       reference="buqeye/gsum v0.3 @ /root/reference"), f, indent=1)


if __name__ == "__main__":
    main()
