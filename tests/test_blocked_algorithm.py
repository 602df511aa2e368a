"""CPU check of the device algorithm's algebra: a numpy emulation of the bordered,
right-looking blocked Cholesky — explicit diagonal-block inverses, border rows -> W^T, corner -> -G —
must reproduce L, W = L^-1 Z, G = W^T W and sum(log diag L) computed directly."""
import numpy as np
from scipy.linalg import solve_triangular

NB, BORDER = 128, 16


def emulate(K, Z):
    n, k = Z.shape
    npad = -(-n // NB) * NB
    T = npad // NB
    A = np.zeros((npad + BORDER, npad + BORDER))
    A[:n, :n] = K
    A[n:npad, n:npad] = np.eye(npad - n)
    A[npad:npad + k, :n] = Z.T
    sld = 0.0
    for s in range(T):
        c0, r0 = s * NB, (s + 1) * NB
        # diag kernel: Cholesky + inverse by forward elimination on an identity
        L11 = np.linalg.cholesky(A[c0:r0, c0:r0])
        Linv = solve_triangular(L11, np.eye(NB), lower=True)
        A[c0:r0, c0:r0] = L11
        sld += np.log(np.diag(L11)).sum()
        # trsm as GEMM against the explicit inverse, rows below (border included)
        A[r0:, c0:r0] = A[r0:, c0:r0] @ Linv.T
        # trailing update (lower part is all that matters; the full update is equivalent)
        P = A[r0:, c0:r0]
        A[r0:, r0:] -= P @ P.T
    L = np.tril(A[:n, :n])
    Wt = A[npad:npad + k, :n]
    G = -A[npad:npad + k, npad:npad + k]
    return L, Wt.T, G, sld


def test_bordered_blocked_cholesky_matches_direct():
    rng = np.random.RandomState(0)
    for n in (5, 128, 200, 391):
        X = 0.1 * np.arange(n)[:, None]          # dx = 0.5 ell: conditioned
        K = np.exp(-0.5 * ((X - X.T) / 0.2) ** 2) + 1e-10 * np.eye(n)
        Z = np.concatenate([rng.randn(n, 4), np.ones((n, 1))], axis=1)
        L, W, G, sld = emulate(K, Z)
        Lr = np.linalg.cholesky(K)
        Wr = solve_triangular(Lr, Z, lower=True)
        # cond(K) ~ 4e7 for this input (lambda_min = 2.7e-8): two backward-stable factorisations agree
        # to ~cond * eps on L / W / G, and far better on the log-likelihood (measured 2e-11 at n = 512)
        np.testing.assert_allclose(L, Lr, rtol=1e-7, atol=1e-9)
        Gr = Wr.T @ Wr
        np.testing.assert_allclose(G, Gr, rtol=2e-8, atol=1e-9 * np.abs(Gr).max())
        s_ref = np.log(np.diag(Lr)).sum()
        assert abs(sld - s_ref) <= 1e-10 * max(1.0, abs(s_ref))
        from gsum_amd.conjugate import lml_from_gram
        got, _ = lml_from_gram(G, sld, n, 0, 0, 1, 1)
        want, _ = lml_from_gram(Wr.T @ Wr, s_ref, n, 0, 0, 1, 1)
        assert abs(got - want) <= 1e-10 * abs(want)


def _potf2_right_looking(A):
    """Unblocked right-looking Cholesky with reciprocal scaling (the order the device's diagonal kernel uses)."""
    A = A.copy()
    n = len(A)
    for j in range(n):
        d = np.sqrt(A[j, j])
        l = A[j + 1:, j] * (1.0 / d)
        A[j, j] = d
        A[j + 1:, j] = l
        A[j + 1:, j + 1:] -= np.outer(l, l)
    return np.tril(A)


def test_intrinsic_spread_of_the_uniform_grid_input():
    """Why tests/test_gpu_parity.py bounds the uniform-grid S-inputs at 3e-10 rather than 1e-10: on the CPU,
    with the SAME kernel matrix, LAPACK's blocked dpotrf and a plain right-looking Cholesky already disagree
    by several 1e-11 in the log-likelihood (cond(K) = 4e7, white-noise coefficients)."""
    from scipy.linalg import solve_triangular
    from gsum_amd.conjugate import lml_from_gram
    n, r = 512, 4
    X = 0.1 * np.arange(n)[:, None]
    K = np.exp(-0.5 * ((X - X.T) / 0.2) ** 2) + 1e-10 * np.eye(n)
    Z = np.concatenate([np.random.RandomState(0).randn(n, r), np.ones((n, 1))], axis=1)
    vals = []
    for chol in (np.linalg.cholesky, _potf2_right_looking):
        L = chol(K)
        W = solve_triangular(L, Z, lower=True)
        vals.append(lml_from_gram(W.T @ W, np.log(np.diag(L)).sum(), n, 0, 0, 1, 1)[0])
    spread = abs(vals[0] - vals[1]) / abs(vals[0])
    assert 1e-12 < spread < 3e-10, spread
