"""Randomised GPU-vs-oracle checks (``-m gpu``): gradients, predictions and truncation predictions on random kernels,
dimensions, priors and sizes, with tolerances scaled by the conditioning of the matrices involved.  Seeds are fixed,
so a failure is reproducible; tools/gpu_fuzz.py does the same for the fused likelihood paths against numpy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import gsum_amd  # noqa: E402
from oracle import gsum_oracle as orc  # noqa: E402  (checker only)
from sklearn.gaussian_process.kernels import RBF, Matern, WhiteKernel, ConstantKernel as C  # noqa: E402


def random_kernel(rng, d, free_white=True):
    fam = rng.choice(["rbf", "m52", "m32", "m12"])
    ls = rng.uniform(0.4, 1.5, size=d) if (d > 1 and rng.rand() < 0.5) else float(rng.uniform(0.4, 1.5))
    base = RBF(ls) if fam == "rbf" else Matern(ls, nu={"m52": 2.5, "m32": 1.5, "m12": 0.5}[fam])
    kern = base
    if rng.rand() < 0.6:
        kern = C(float(rng.uniform(0.5, 2.5))) * kern
    w = float(10 ** rng.uniform(-6, -2))
    kern = kern + (WhiteKernel(w) if (free_white and rng.rand() < 0.5) else WhiteKernel(w, noise_level_bounds="fixed"))
    if rng.rand() < 0.3:
        kern = kern + C(float(rng.uniform(0.05, 0.5)))
    return kern


def random_priors(rng):
    p = dict(center=float(rng.uniform(-0.5, 0.5)), disp=float(rng.choice([0.0, rng.uniform(0.3, 2.0)])))
    if rng.rand() < 0.25:
        p["sd"] = float(rng.uniform(0.6, 1.8))
    else:
        p["df"] = float(rng.uniform(1.0, 6.0))
        p["scale"] = float(rng.uniform(0.6, 1.8))
    return p


def drawn(rng, kern, X, cols):
    L = np.linalg.cholesky(kern(X) + 1e-8 * np.eye(len(X)))
    return 0.2 + L @ rng.randn(len(X), cols)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_lml_gradient_vs_oracle(seed):
    rng = np.random.RandomState(1000 + seed)
    d = int(rng.randint(1, 4))
    n = int(rng.choice([9, 33, 128, 129, 200, 257, 400]))
    kern = random_kernel(rng, d)
    X = rng.rand(n, d) * (3.0 + 0.02 * n)
    y = drawn(rng, kern, X, int(rng.randint(1, 5)))
    pri = random_priors(rng)
    theta = kern.theta + rng.uniform(-0.2, 0.2, size=len(kern.theta))
    cond = np.linalg.cond(kern.clone_with_theta(theta)(X) + 1e-10 * np.eye(n))
    for cls, ofn in ((gsum_amd.ConjugateGaussianProcess, orc.cgp_lml_grad), (gsum_amd.ConjugateStudentProcess, orc.csp_lml_grad)):
        if cls is gsum_amd.ConjugateStudentProcess and "sd" in pri:
            continue                                    # df0 = inf: inf - inf in the reference's Student normalisation
        gp = cls(kernel=kern, optimizer=None, **pri)
        val, grad = gp.log_marginal_likelihood(theta, eval_gradient=True, X=X, y=y)
        vo, go = ofn(kern, theta, X, y, **pri)
        assert val == pytest.approx(vo, rel=max(1e-10, 1e-16 * cond))
        tol = 1e-14 * cond + 1e-9
        np.testing.assert_allclose(grad, go, rtol=tol, atol=tol * np.abs(go).max())


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_fit_predict_vs_oracle(seed):
    rng = np.random.RandomState(2000 + seed)
    d = int(rng.randint(1, 4))
    n = int(rng.choice([7, 64, 128, 130, 300, 700]))
    m = int(rng.choice([1, 5, 40, 150]))
    kern = random_kernel(rng, d, free_white=False)
    X, Xs = rng.rand(n, d) * (3.0 + 0.02 * n), rng.rand(m, d) * (3.0 + 0.02 * n)
    y = drawn(rng, kern, X, int(rng.randint(1, 4)))
    pri = random_priors(rng)
    gp = gsum_amd.ConjugateGaussianProcess(kernel=kern, optimizer=None, **pri).fit(X, y)
    fit = orc.cgp_fit(kern, X, y, **pri)
    cond = np.linalg.cond(fit["corr"] + 1e-10 * np.eye(n))
    tol = 1e-14 * cond + 1e-10
    np.testing.assert_allclose(gp.center_, fit["center"], rtol=tol, atol=tol)
    assert gp.cov_factor_ == pytest.approx(fit["cov_factor"], rel=tol)
    mo, co = orc.cgp_predict(fit, Xs, return_cov=True)
    mg, cg = gp.predict(Xs, return_cov=True)
    np.testing.assert_allclose(mg, mo, rtol=tol, atol=tol * max(1.0, np.abs(mo).max()))
    np.testing.assert_allclose(cg, co, rtol=1e-6, atol=tol * fit["cov_factor"])
    sg = gp.predict(Xs, return_std=True)[1]
    if m > 1:           # with one new point the reference (and the oracle) squeeze the 1 x 1 covariance and np.diag raises
        so = orc.cgp_predict(fit, Xs, return_std=True)[1]
        np.testing.assert_allclose(sg ** 2, so ** 2, rtol=1e-6, atol=tol * fit["cov_factor"])
    else:
        np.testing.assert_allclose(np.atleast_1d(sg) ** 2, np.atleast_1d(np.diag(np.atleast_2d(co))), rtol=1e-6,
                                   atol=tol * fit["cov_factor"])


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_truncation_predict_vs_oracle(seed):
    rng = np.random.RandomState(3000 + seed)
    n = int(rng.choice([6, 10, 40, 150]))
    X = np.sort(rng.rand(n))[:, None] * (0.8 * n)             # ~0.8 apart on average: a well-conditioned K_oo
    Xs = rng.rand(int(rng.choice([3, 20])), 1) * (0.8 * n)
    fam = rng.choice(["rbf", "m52"])
    base = RBF(float(rng.uniform(0.5, 0.9))) if fam == "rbf" else Matern(float(rng.uniform(0.6, 1.2)), nu=2.5)
    kern = C(float(rng.uniform(0.5, 2.0))) * base + WhiteKernel(1e-6, noise_level_bounds="fixed")
    orders = np.arange(int(rng.randint(3, 6)))
    excluded = [1] if rng.rand() < 0.4 else None
    order = int(orders[-2])
    if rng.rand() < 0.5:
        a, b = rng.uniform(0.25, 0.45), rng.uniform(0.0, 0.15)
        ratio = lambda Xa, a=a, b=b: a + b * np.sin(Xa[:, 0]) ** 2       # noqa: E731
        ref = lambda Xa: 1.0 + 0.01 * Xa[:, 0]                           # noqa: E731
    else:
        ratio, ref = float(rng.uniform(0.3, 0.6)), float(rng.uniform(0.5, 3.0))
    rv = ratio(X) if callable(ratio) else ratio
    fv = ref(X) if callable(ref) else ref
    c = drawn(rng, kern, X, len(orders))
    y = gsum_amd.partials(c, ratio=rv, ref=fv, orders=orders)
    pri = dict(center=0.0, disp=0.0, df=3.0, scale=1.0)
    gp = gsum_amd.TruncationGP(kernel=kern, ratio=ratio, ref=ref, excluded=excluded, optimizer=None, **pri)
    gp.fit(X, y, orders=orders)
    cc = orc.coefficients(y, rv, fv, orders)[:, ~np.isin(orders, excluded)]
    fit = orc.cgp_fit(kern, X, cc, **pri)
    yo = y[:, orders == order][:, 0]
    K_oo = orc.trunc_cov(fit["cov_factor"], kern, X, X, ratio, ref, 0, order, excluded)
    tol = 1e-13 * np.linalg.cond(K_oo) + 1e-10
    for kind in ("interp", "both", "trunc"):
        mo, co = orc.trunc_predict(fit["center"], fit["cov_factor"], kern, Xs, order, ratio, ref, X, yo, excluded=excluded,
                                   kind=kind, return_cov=True)
        mg, cg = gp.predict(Xs, order=order, return_cov=True, kind=kind)
        np.testing.assert_allclose(mg, mo, rtol=tol, atol=tol * np.abs(mo).max())
        np.testing.assert_allclose(cg, co, rtol=1e-6, atol=tol * max(np.abs(co).max(), 1e-300))


def random_tree(rng, d):
    """A random Sum / Product / Exponentiation tree over every leaf class the device evaluates (round 5: ExpSineSquared, Matern(nu = inf),
    DotProduct; round 4: RationalQuadratic), positive definite by construction (products and integer powers of positive definite kernels,
    a white-noise floor)."""
    from sklearn.gaussian_process.kernels import DotProduct, Exponentiation, ExpSineSquared, RationalQuadratic

    def leaf():
        kind = rng.choice(["rbf", "m52", "m32", "m12", "minf", "rq", "ess", "dot"] if d == 1 else ["rbf", "m52", "m32", "m12", "minf", "rq", "dot"])
        ls = rng.uniform(0.5, 1.6, size=d) if (d > 1 and rng.rand() < 0.5 and kind not in ("rq", "ess", "dot")) else float(rng.uniform(0.5, 1.6))
        if kind == "rbf":
            return RBF(ls)
        if kind == "rq":
            return RationalQuadratic(length_scale=ls, alpha=float(rng.uniform(0.5, 2.0)))
        if kind == "ess":                     # (a periodic kernel of the Euclidean distance is positive definite in one dimension only)
            return ExpSineSquared(length_scale=ls, periodicity=float(rng.uniform(2.0, 5.0)))
        if kind == "dot":
            return C(0.05) * DotProduct(sigma_0=float(rng.uniform(0.5, 2.0)))
        return Matern(ls, nu={"m52": 2.5, "m32": 1.5, "m12": 0.5, "minf": np.inf}[kind])

    n_leaves = int(rng.randint(1, 4))
    kern = None
    for _ in range(n_leaves):
        term = leaf()
        if rng.rand() < 0.4:
            term = C(float(rng.uniform(0.5, 2.0))) * term
        if rng.rand() < 0.25:
            term = Exponentiation(term, 2)
        kern = term if kern is None else (kern * term if rng.rand() < 0.4 else kern + term)
    w = float(10 ** rng.uniform(-4, -2))
    return kern + (WhiteKernel(w) if rng.rand() < 0.5 else WhiteKernel(w, noise_level_bounds="fixed"))


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_kernel_trees_vs_oracle(seed):
    """Random kernel trees (models.py:146-147, 958-960 accept any scikit-learn kernel): the matrix against scikit-learn's, likelihood and
    gradient of both process classes and the predictive mean / standard deviation against the oracle; tolerances scaled by conditioning."""
    rng = np.random.RandomState(7000 + seed)
    d = int(rng.randint(1, 3))
    n = int(rng.choice([17, 60, 128, 150, 300]))
    while True:
        kern = random_tree(rng, d)
        try:
            desc = gsum_amd.describe_kernel(kern, d)
            break
        except NotImplementedError as exc:                # (a draw beyond the descriptor's 24 operations: the next one)
            assert "too large" in str(exc)
    check_tree_kernel(kern, desc, rng, n, d)


def check_tree_kernel(kern, desc, rng, n, d):
    X = rng.rand(n, d) * (3.0 + 0.02 * n)
    Xs = rng.rand(11, d) * (3.0 + 0.02 * n)
    ctx = gsum_amd.default_context(0)
    K, Kx = ctx.kernel_matrix(desc, X), ctx.kernel_matrix(desc, X, Xs)
    scale = np.abs(kern(X)).max()
    np.testing.assert_allclose(K, kern(X), rtol=1e-13, atol=1e-14 * scale)
    np.testing.assert_allclose(Kx, kern(X, Xs), rtol=1e-13, atol=1e-14 * scale)
    y = drawn(rng, kern, X, int(rng.randint(1, 4)))
    pri = random_priors(rng)
    theta = kern.theta + rng.uniform(-0.15, 0.15, size=len(kern.theta))
    cond = np.linalg.cond(kern.clone_with_theta(theta)(X) + 1e-10 * np.eye(n))
    for cls, ofn in ((gsum_amd.ConjugateGaussianProcess, orc.cgp_lml_grad), (gsum_amd.ConjugateStudentProcess, orc.csp_lml_grad)):
        if cls is gsum_amd.ConjugateStudentProcess and "sd" in pri:
            continue
        gp = cls(kernel=kern, optimizer=None, **pri)
        val, grad = gp.log_marginal_likelihood(theta, eval_gradient=True, X=X, y=y)
        vo, go = ofn(kern, theta, X, y, **pri)
        assert val == pytest.approx(vo, rel=max(1e-10, 1e-15 * cond)), kern
        tol = 1e-14 * cond + 1e-9
        np.testing.assert_allclose(grad, go, rtol=tol, atol=tol * np.abs(go).max(), err_msg=str(kern))
    gp = gsum_amd.ConjugateGaussianProcess(kernel=kern, optimizer=None, **pri)
    gp.fit(X, y)
    mean, std = gp.predict(Xs, return_std=True)
    fit = orc.cgp_fit(kern, X, y, **pri)
    condf = np.linalg.cond(fit["corr"] + 1e-10 * np.eye(n))
    ptol = 1e-14 * condf + 1e-10
    mo, so = orc.cgp_predict(fit, Xs, return_std=True)
    np.testing.assert_allclose(mean, mo, rtol=ptol, atol=ptol * max(1.0, np.abs(mo).max()), err_msg=str(kern))
    np.testing.assert_allclose(std ** 2, so ** 2, rtol=1e-6, atol=ptol * fit["cov_factor"] * max(1.0, scale), err_msg=str(kern))


@pytest.mark.parametrize("n", [40, 128, 300, 1100])
def test_four_scaled_leaves_and_white_noise(n):
    """The largest sum the descriptor holds: four leaves of four families, each with its ConstantKernel factor, plus WhiteKernel -- 17 operations, ten
    hyperparameters (the descriptor took 16 operations until late round 5) -- on the one-block, general and one-workgroup paths; and one leaf more is
    refused on the host with the reason."""
    from sklearn.gaussian_process.kernels import ConstantKernel as C, Matern, RBF, RationalQuadratic, WhiteKernel
    rng = np.random.RandomState(n)
    kern = (C(1.2) * RBF(0.9) + C(0.4) * Matern(1.7, nu=1.5) + C(0.8) * Matern(0.6, nu=2.5) + C(0.3) * RationalQuadratic(length_scale=1.1, alpha=0.8)
            + WhiteKernel(1e-3))
    desc = gsum_amd.describe_kernel(kern, 1)
    assert desc.n_ops == 17 and desc.n_leaves == 4 and len(kern.theta) == 10
    check_tree_kernel(kern, desc, rng, n, 1)
    # a surface over it (n = 300, 1100: 12 evaluations take the one-workgroup-per-evaluation path at 300) equals the single calls
    X = np.sort(rng.rand(n))[:, None] * (3.0 + 0.02 * n)
    y = drawn(rng, kern, X, 2)
    gp = gsum_amd.ConjugateGaussianProcess(kernel=kern, optimizer=None, center=0, disp=0, df=1, scale=1)
    thetas = [kern.theta + 0.02 * j for j in range(12)]
    batch = gp.log_marginal_likelihood_batch(thetas, X=X, y=y)
    for j in (0, 5, 11):
        one = gp.log_marginal_likelihood(thetas[j], eval_gradient=True, X=X, y=y)
        assert batch[j][0] == one[0] and np.array_equal(batch[j][1], one[1])
    with pytest.raises(NotImplementedError, match="too large"):
        gsum_amd.describe_kernel(kern + C(0.1) * RBF(3.0), 1)
