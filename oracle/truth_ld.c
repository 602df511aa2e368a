/* truth_ld.c -- extended-precision (x87 80-bit long double, 64-bit significand) evaluation of the pieces the
 * log-likelihood of gsum/models.py:958-1039 is made of, on a matrix given in fp64.  TEST INFRASTRUCTURE ONLY
 * (tests/golden/make_truth.py writes fixtures with it; nothing under gsum_amd/ links or loads it).
 *
 * Given the fp64 matrix R (the bit pattern scikit-learn's kernel(X) + nugget produces -- the device kernel build is
 * asserted array_equal to it) and right-hand sides Z (n x k, fp64), it computes in long double
 *     L = chol(R)  (numpy.linalg.cholesky, models.py:969),  W = L^-1 Z,  G = W^T W,  s = sum_i log L_ii
 * i.e. the exact-arithmetic targets of cho_solve / einsum / log-det at models.py:1015, 1032-1035, to ~2^-64 relative
 * rounding per operation instead of 2^-53: the error of this evaluation is ~2000x below that of any fp64
 * factorisation, which makes it a yardstick for "how far is LAPACK from the true value, how far is the HIP path".
 * Blocked right-looking factorisation (64-column panels), OpenMP over row blocks; 16 bytes per entry.
 * Build: gcc -O2 -fopenmp -shared -fPIC -o oracle/_build/libtruth_ld.so oracle/truth_ld.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef long double ld;
#define NB 64

/* C[i][j] -= sum_k A[i][k] B[j][k], i < mi, j < nj, k < kk; row strides lda/ldb/ldc */
static void gemm_nt_sub(ld* C, int64_t ldc, const ld* A, int64_t lda, const ld* B, int64_t ldb, int mi, int nj, int kk, int lower_only,
                        int i0, int j0) {
    for (int i = 0; i < mi; ++i)
        for (int j = 0; j < nj; ++j) {
            if (lower_only && j0 + j > i0 + i) break;
            const ld* a = A + (int64_t)i * lda;
            const ld* b = B + (int64_t)j * ldb;
            ld s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            int k = 0;
            for (; k + 3 < kk; k += 4) {
                s0 += a[k] * b[k];
                s1 += a[k + 1] * b[k + 1];
                s2 += a[k + 2] * b[k + 2];
                s3 += a[k + 3] * b[k + 3];
            }
            for (; k < kk; ++k) s0 += a[k] * b[k];
            C[(int64_t)i * ldc + j] -= (s0 + s1) + (s2 + s3);
        }
}

/* returns 0, or the 1-based index of the first non-positive pivot.  G: k x k, sld: 1, both as long double written
 * into caller buffers of 16-byte elements (numpy.longdouble). */
int truth_gram_ld(const double* R, const double* Z, int64_t n, int k, ld* G, ld* sld) {
    const int64_t ldw = n;
    ld* A = (ld*)malloc((size_t)n * n * sizeof(ld));
    ld* W = (ld*)malloc((size_t)n * k * sizeof(ld));
    if (!A || !W) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        for (int64_t j = 0; j <= i; ++j) A[i * ldw + j] = (ld)R[i * n + j];
        for (int c = 0; c < k; ++c) W[i * k + c] = (ld)Z[i * k + c];
    }
    int info = 0;
    for (int64_t c0 = 0; c0 < n && !info; c0 += NB) {
        const int nb = (int)((n - c0) < NB ? (n - c0) : NB);
        /* diagonal block: unblocked */
        for (int j = 0; j < nb && !info; ++j) {
            ld* row = A + (c0 + j) * ldw + c0;
            ld d = row[j];
            for (int p = 0; p < j; ++p) d -= row[p] * row[p];
            if (!(d > 0)) {
                info = (int)(c0 + j + 1);
                break;
            }
            d = sqrtl(d);
            row[j] = d;
            for (int i = j + 1; i < nb; ++i) {
                ld* ri = A + (c0 + i) * ldw + c0;
                ld s = ri[j];
                for (int p = 0; p < j; ++p) s -= ri[p] * row[p];
                ri[j] = s / d;
            }
        }
        if (info) break;
        const int64_t r0 = c0 + nb;
        /* panel: rows below, forward substitution against the diagonal block */
#pragma omp parallel for schedule(dynamic, 16)
        for (int64_t i = r0; i < n; ++i) {
            ld* ri = A + i * ldw + c0;
            for (int j = 0; j < nb; ++j) {
                const ld* rj = A + (c0 + j) * ldw + c0;
                ld s = ri[j];
                for (int p = 0; p < j; ++p) s -= ri[p] * rj[p];
                ri[j] = s / rj[j];
            }
        }
        /* trailing update, lower tiles */
        const int64_t nt = (n - r0 + NB - 1) / NB;
#pragma omp parallel for schedule(dynamic, 1)
        for (int64_t t = 0; t < nt * nt; ++t) {
            const int64_t bi = t / nt, bj = t % nt;
            if (bj > bi) continue;
            const int64_t i0 = r0 + bi * NB, j0 = r0 + bj * NB;
            const int mi = (int)((n - i0) < NB ? (n - i0) : NB), nj = (int)((n - j0) < NB ? (n - j0) : NB);
            gemm_nt_sub(A + i0 * ldw + j0, ldw, A + i0 * ldw + c0, ldw, A + j0 * ldw + c0, ldw, mi, nj, nb, bi == bj, (int)i0, (int)j0);
        }
    }
    if (!info) {
        /* W = L^-1 Z (column by column in parallel), G = W^T W, s = sum log L_ii */
#pragma omp parallel for schedule(static)
        for (int c = 0; c < k; ++c)
            for (int64_t i = 0; i < n; ++i) {
                const ld* li = A + i * ldw;
                ld s0 = 0, s1 = 0;
                int64_t p = 0;
                for (; p + 1 < i; p += 2) {
                    s0 += li[p] * W[p * k + c];
                    s1 += li[p + 1] * W[(p + 1) * k + c];
                }
                for (; p < i; ++p) s0 += li[p] * W[p * k + c];
                W[i * k + c] = (W[i * k + c] - (s0 + s1)) / li[i];
            }
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) {
                ld s = 0;
                for (int64_t i = 0; i < n; ++i) s += W[i * k + a] * W[i * k + b];
                G[a * k + b] = s;
            }
        ld s = 0;
        for (int64_t i = 0; i < n; ++i) s += logl(A[i * ldw + i]);
        *sld = s;
    }
    free(A);
    free(W);
    return info;
}
