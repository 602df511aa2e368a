// Device kernels of libgsum_hip.so — written for gfx950 (MI355X, CDNA4) only.
//
//   k_build        pairwise-distance + RBF/Matern kernel matrix   (HBM-store bound)
//   k_set_border   RHS^T -> border rows of the augmented matrix
//   k_potrf_diag   128x128 diagonal block: Cholesky + inverse, in registers
//   k_gemm_nt      C (+)= s * A * B^T on v_mfma_f64_16x16x4_f64   (fp64 MFMA bound)
//   k_finalize     Gram / log-det read-out of the bordered factorisation
//   k_rowsumsq     row-wise sum of squares (predictive variance)
//   probes         fp64 MFMA issue rate, HBM store rate
//
// Data layout: the factorisation works on ONE augmented row-major fp64 matrix
//     [ K (np x np, lower triangle)  .            ]      np = n rounded up to 128 (identity padding)
//     [ RHS^T (16 x np)              -G (16 x 16) ]      leading dimension ld = np + 16
// A right-looking blocked Cholesky over the first np columns turns the border rows into
// W^T = (L^-1 RHS)^T and the corner into -W^T W, so the forward solve and the Gram reduction of the
// log-likelihood cost no extra pass over L.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gsum_hip.h"

#define GS_NB 128
#define GS_BORDER 16
#define GS_KC 16                  // K chunk staged through LDS (16 doubles = one 128-B line per row)
#define GS_LSTR (GS_KC + 2)       // padded LDS row stride: conflict-free ds_read_b64 fragment reads

typedef double gs_d4 __attribute__((ext_vector_type(4)));
typedef double gs_d2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// K1: kernel matrix build
// ------------------------------------------------------------------------------------------------
// sklearn arithmetic, restated (sklearn/gaussian_process/kernels.py):
//   RBF     :1556-1560  exp(-0.5 * sqeuclidean(X/ls)), diagonal forced to 1
//   Matern  :1711-1738  d = euclidean(X/ls); nu=.5: exp(-d); 1.5: t=d*sqrt3, (1+t)exp(-t);
//                       2.5: t=d*sqrt5, (1+t+t*t/3)exp(-t)
//   Product :956-966 (amplitude * base), Sum :858-866 (+ constant), White :1401-1414 (+ noise on diag)
// Floating-point contraction is off so that sums of squares round like the host code does.
__device__ __forceinline__ double gs_base_value(int family, double s) {
#pragma clang fp contract(off)
    if (family == GSUM_RBF) return exp(-0.5 * s);
    double dist = sqrt(s);
    if (family == GSUM_MATERN52) {
        double t = dist * 2.23606797749979;      // math.sqrt(5)
        return (1.0 + t + (t * t) / 3.0) * exp(-t);
    }
    if (family == GSUM_MATERN32) {
        double t = dist * 1.7320508075688772;    // math.sqrt(3)
        return (1.0 + t) * exp(-t);
    }
    return exp(-dist);
}

// One 128x128 tile per 256-thread workgroup.  Each lane owns two adjacent columns (one 16-B store per
// row), each wave strides over the tile's rows: every store instruction writes 1 KiB of one row.
// CROSS=false: symmetric one-argument form into the (identity-padded) square matrix; tri!=0 builds
// only tiles on or below the diagonal.  CROSS=true: rectangular k(X, Y), no diagonal terms.
template <bool CROSS>
__global__ __launch_bounds__(256) void k_build(double* out, int64_t ldo, const double* X, const double* Y,
                                                int n, int m, int prow, int pcol, int d,
                                                gsum_kernel_desc desc, double diag_add, int tri) {
#pragma clang fp contract(off)
    __shared__ double ui[128 * GSUM_MAX_D];
    __shared__ double uj[128 * GSUM_MAX_D];
    const int t = threadIdx.x;
    int bi, bj;
    if (tri) {
        int bid = blockIdx.x;
        bi = (int)((sqrt(8.0 * (double)bid + 1.0) - 1.0) * 0.5);
        while ((int64_t)(bi + 1) * (bi + 2) / 2 <= bid) ++bi;
        while ((int64_t)bi * (bi + 1) / 2 > bid) --bi;
        bj = bid - (int)((int64_t)bi * (bi + 1) / 2);
    } else {
        int tr = (prow + 127) / 128;
        bi = blockIdx.x % tr;
        bj = blockIdx.x / tr;
    }
    const double* Yp = CROSS ? Y : X;
    const int ny = CROSS ? m : n;
    for (int idx = t; idx < 128 * d; idx += 256) {
        int r = idx / d, dd = idx - r * d;
        double ls = desc.anisotropic ? desc.length_scale[dd] : desc.length_scale[0];
        int gi = bi * 128 + r, gj = bj * 128 + r;
        ui[idx] = gi < n ? X[(int64_t)gi * d + dd] / ls : 0.0;
        uj[idx] = gj < ny ? Yp[(int64_t)gj * d + dd] / ls : 0.0;
    }
    __syncthreads();
    const int lane = t & 63, w = t >> 6;
    const int gj0 = bj * 128 + 2 * lane;
    double vj0[GSUM_MAX_D], vj1[GSUM_MAX_D];
#pragma unroll
    for (int dd = 0; dd < GSUM_MAX_D; ++dd) {
        vj0[dd] = dd < d ? uj[(2 * lane) * d + dd] : 0.0;
        vj1[dd] = dd < d ? uj[(2 * lane + 1) * d + dd] : 0.0;
    }
    if (gj0 >= pcol) return;
    for (int rr = w; rr < 128; rr += 4) {
        const int gi = bi * 128 + rr;
        if (gi >= prow) break;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int dd = 0; dd < GSUM_MAX_D; ++dd) {
            if (dd < d) {
                double xi = ui[rr * d + dd];
                double e0 = xi - vj0[dd], e1 = xi - vj1[dd];
                s0 = s0 + e0 * e0;
                s1 = s1 + e1 * e1;
            }
        }
        double v[2];
        const double s[2] = {s0, s1};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int gj = gj0 + c;
            double val;
            if (CROSS) {
                val = (gi < n && gj < m) ? desc.amplitude * gs_base_value(desc.family, s[c]) + desc.additive_const : 0.0;
            } else if (gi >= n || gj >= n) {
                val = (gi == gj) ? 1.0 : 0.0;                 // identity padding up to a multiple of 128
            } else {
                const bool dg = gi == gj;
                double b = dg ? 1.0 : gs_base_value(desc.family, s[c]);   // np.fill_diagonal(K, 1)
                val = desc.amplitude * b;
                if (dg) val = val + desc.white_noise;
                val = val + desc.additive_const;
                if (dg) val = val + diag_add;
            }
            v[c] = val;
        }
        gs_d2 o = {v[0], v[1]};
        *reinterpret_cast<gs_d2*>(out + (int64_t)gi * ldo + gj0) = o;
    }
}

// Border rows np..np+15 of the augmented matrix: row c = column c of RHS (n x k, row-major), zero
// beyond k / n, and a zero 16x16 corner.
__global__ __launch_bounds__(256) void k_set_border(double* A, int64_t ld, int n, int np, const double* Z, int k) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= np + GS_BORDER) return;
#pragma unroll
    for (int c = 0; c < GS_BORDER; ++c)
        A[(int64_t)(np + c) * ld + i] = (c < k && i < n) ? Z[(int64_t)i * k + c] : 0.0;
}

// Rows >= n of the padded square part become identity rows; used after a host upload.
__global__ __launch_bounds__(256) void k_pad_identity(double* A, int64_t ld, int n, int np) {
    int j = blockIdx.x * 256 + threadIdx.x;
    int i = n + blockIdx.y;
    if (j >= np || i >= np) return;
    A[(int64_t)i * ld + j] = (i == j) ? 1.0 : 0.0;
}

// ------------------------------------------------------------------------------------------------
// K2a: diagonal block — unblocked right-looking Cholesky of a 128x128 block held ENTIRELY IN
// REGISTERS (2-D cyclic 16x16 thread grid, 36 lower 16x16 sub-blocks -> 36 doubles per thread), with
// the inverse of the factor built alongside by forward elimination on an identity (another 36).
// Each column costs one barrier: the unscaled column and the inverse's pivot row go through a
// double-buffered 2 KiB LDS mailbox.  LAPACK dpotf2 semantics: pivot <= 0 or NaN -> info.
// ------------------------------------------------------------------------------------------------
template <int JB>
__device__ __forceinline__ bool gs_diag_steps(double (&a)[8][8], double (&v)[8][8], double* colbuf,
                                              double* rowbuf, double* dbuf, int tr, int tc, int* fail_col) {
    for (int jr = 0; jr < 16; ++jr) {
        const int j = JB * 16 + jr;
        double* cb = colbuf + (j & 1) * 128;
        double* rb = rowbuf + (j & 1) * 128;
        if (tc == jr) {
#pragma unroll
            for (int ii = JB; ii < 8; ++ii) cb[tr + 16 * ii] = a[ii][JB];
        }
        if (tr == jr) {
#pragma unroll
            for (int kk = 0; kk <= JB; ++kk) rb[tc + 16 * kk] = v[JB][kk];
        }
        __syncthreads();
        const double p = cb[j];
        if (!(p > 0.0)) {           // same value in every thread: uniform exit
            *fail_col = j;
            return false;
        }
        const double dj = sqrt(p);
        const double r = 1.0 / dj;
        if (threadIdx.x == 0) dbuf[j] = dj;
        double li[8], lk[8], vk[8];
#pragma unroll
        for (int ii = JB; ii < 8; ++ii) {
            const int row = tr + 16 * ii;
            li[ii] = (row > j) ? cb[row] * r : 0.0;
        }
#pragma unroll
        for (int kk = JB; kk < 8; ++kk) {
            const int col = tc + 16 * kk;
            lk[kk] = (col > j) ? cb[col] * r : 0.0;
        }
#pragma unroll
        for (int kk = 0; kk <= JB; ++kk) {
            const int col = tc + 16 * kk;
            vk[kk] = (col <= j) ? rb[col] * r : 0.0;
        }
        // trailing update of the block:  A_ik -= l_ij l_kj   (i, k > j)
#pragma unroll
        for (int ii = JB; ii < 8; ++ii)
#pragma unroll
            for (int kk = JB; kk <= ii; ++kk) a[ii][kk] -= li[ii] * lk[kk];
        // column j is final: l_ij below the diagonal, d_j on it
        if (tc == jr) {
#pragma unroll
            for (int ii = JB; ii < 8; ++ii) {
                const int row = tr + 16 * ii;
                a[ii][JB] = (row > j) ? li[ii] : ((row == j) ? dj : a[ii][JB]);
            }
        }
        // inverse: rows below j eliminate against the scaled pivot row
#pragma unroll
        for (int ii = JB; ii < 8; ++ii)
#pragma unroll
            for (int kk = 0; kk <= JB; ++kk) v[ii][kk] -= li[ii] * vk[kk];
        if (tr == jr) {
#pragma unroll
            for (int kk = 0; kk <= JB; ++kk) v[JB][kk] = vk[kk];
        }
    }
    return true;
}

// A: pointer to the diagonal block inside the augmented matrix (leading dimension ld).
// Linv: 128x128 row-major output (zeros above the diagonal).  logdet[0] = sum_j log L_jj.
// info: global failure flag (0 = ok so far; >0 = LAPACK-style 1-based failing column).
__global__ __launch_bounds__(256) void k_potrf_diag(double* A, int64_t ld, double* Linv, double* logdet,
                                                     int* info, int col0) {
    __shared__ double colbuf[256];
    __shared__ double rowbuf[256];
    __shared__ double dbuf[128];
    if (*info != 0) return;                    // an earlier block already failed (uniform)
    const int t = threadIdx.x;
    const int tr = t >> 4, tc = t & 15;
    double a[8][8], v[8][8];
#pragma unroll
    for (int ii = 0; ii < 8; ++ii)
#pragma unroll
        for (int kk = 0; kk <= ii; ++kk) {
            a[ii][kk] = A[(int64_t)(tr + 16 * ii) * ld + tc + 16 * kk];
            v[ii][kk] = (ii == kk && tr == tc) ? 1.0 : 0.0;
        }
    int fail_col = -1;
    bool ok = gs_diag_steps<0>(a, v, colbuf, rowbuf, dbuf, tr, tc, &fail_col);
    if (ok) ok = gs_diag_steps<1>(a, v, colbuf, rowbuf, dbuf, tr, tc, &fail_col);
    if (ok) ok = gs_diag_steps<2>(a, v, colbuf, rowbuf, dbuf, tr, tc, &fail_col);
    if (ok) ok = gs_diag_steps<3>(a, v, colbuf, rowbuf, dbuf, tr, tc, &fail_col);
    if (ok) ok = gs_diag_steps<4>(a, v, colbuf, rowbuf, dbuf, tr, tc, &fail_col);
    if (ok) ok = gs_diag_steps<5>(a, v, colbuf, rowbuf, dbuf, tr, tc, &fail_col);
    if (ok) ok = gs_diag_steps<6>(a, v, colbuf, rowbuf, dbuf, tr, tc, &fail_col);
    if (ok) ok = gs_diag_steps<7>(a, v, colbuf, rowbuf, dbuf, tr, tc, &fail_col);
    if (!ok) {
        if (t == 0) *info = col0 + fail_col + 1;
        return;
    }
#pragma unroll
    for (int ii = 0; ii < 8; ++ii)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int row = tr + 16 * ii, col = tc + 16 * kk;
            if (kk <= ii) {
                if (col <= row) A[(int64_t)row * ld + col] = a[ii][kk];
                Linv[row * 128 + col] = (col <= row) ? v[ii][kk] : 0.0;
            } else {
                Linv[row * 128 + col] = 0.0;
            }
        }
    __syncthreads();
    if (t < 128) dbuf[t] = log(dbuf[t]);
    __syncthreads();
    if (t == 0) {
        double s = 0.0;
        for (int j = 0; j < 128; ++j) s += dbuf[j];
        logdet[0] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// K2b: C (+)= sign * A * B^T  on fp64 MFMA.   A: M x K, B: N x K (both row-major, K contiguous — the
// shape every step of a row-major lower Cholesky produces), C: M x N.
//   - 4 waves per workgroup, wave tile (WM*16) x (WN*16) of v_mfma_f64_16x16x4_f64 accumulators;
//   - K is staged 16 doubles (one 128-B line per row) at a time: global -> registers -> LDS, two LDS
//     stages, one barrier per chunk; the next chunk's global loads are in flight during the MFMAs;
//   - fragment reads are ds_read_b64 at row stride 18 doubles: conflict-free for the A/B lane map
//     (lane l holds [row l&15][k l>>4]);
//   - rows >= M / cols >= N are clamped on load and predicated on store, so the 16-row border tile
//     and the padded tail run through the same code;
//   - tri != 0: only tiles on or below the diagonal (SYRK of the trailing matrix).
// In-place use (C == A, TRSM against an explicit inverse) is safe when one tile spans all N = K
// columns: every global load of the tile's rows is finished before the epilogue stores.
// ------------------------------------------------------------------------------------------------
template <int WM, int WN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256, 2) void k_gemm_nt(double* C, int64_t ldc, const double* A, int64_t lda,
                                                     const double* B, int64_t ldb, int M, int N, int K,
                                                     int tri, int beta, double sign) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int BM = WM * 16 * WAVES_M, BN = WN * 16 * WAVES_N;
    constexpr int A_VECS = BM * (GS_KC / 2), B_VECS = BN * (GS_KC / 2);
    constexpr int A_IT = (A_VECS + 255) / 256, B_IT = (B_VECS + 255) / 256;
    extern __shared__ double lds[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w % WAVES_M, wn = w / WAVES_M;
    int bm, bn;
    if (tri) {
        const int bid = blockIdx.x;
        bm = (int)((sqrt(8.0 * (double)bid + 1.0) - 1.0) * 0.5);
        while ((int64_t)(bm + 1) * (bm + 2) / 2 <= bid) ++bm;
        while ((int64_t)bm * (bm + 1) / 2 > bid) --bm;
        bn = bid - (int)((int64_t)bm * (bm + 1) / 2);
    } else {
        const int tm = (M + BM - 1) / BM;
        bm = blockIdx.x % tm;
        bn = blockIdx.x / tm;
    }
    const int m0 = bm * BM, n0 = bn * BN;

    gs_d4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = (gs_d4){0.0, 0.0, 0.0, 0.0};

    gs_d2 ra[A_IT], rb[B_IT];
    auto gload = [&](int kc) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int vv = t + it * 256;
            if (vv < A_VECS) {
                int row = m0 + (vv >> 3);
                row = row < M ? row : M - 1;
                ra[it] = *reinterpret_cast<const gs_d2*>(A + (int64_t)row * lda + kc * GS_KC + 2 * (vv & 7));
            }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int vv = t + it * 256;
            if (vv < B_VECS) {
                int row = n0 + (vv >> 3);
                row = row < N ? row : N - 1;
                rb[it] = *reinterpret_cast<const gs_d2*>(B + (int64_t)row * ldb + kc * GS_KC + 2 * (vv & 7));
            }
        }
    };
    auto swrite = [&](int stage) {
        double* sA = lds + stage * (BM + BN) * GS_LSTR;
        double* sB = sA + BM * GS_LSTR;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int vv = t + it * 256;
            if (vv < A_VECS) *reinterpret_cast<gs_d2*>(sA + (vv >> 3) * GS_LSTR + 2 * (vv & 7)) = ra[it];
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int vv = t + it * 256;
            if (vv < B_VECS) *reinterpret_cast<gs_d2*>(sB + (vv >> 3) * GS_LSTR + 2 * (vv & 7)) = rb[it];
        }
    };

    const int nk = K / GS_KC;
    gload(0);
    swrite(0);
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    for (int c = 0; c < nk; ++c) {
        if (c + 1 < nk) gload(c + 1);
        const double* sA = lds + (c & 1) * (BM + BN) * GS_LSTR + (wm * WM * 16 + fr) * GS_LSTR + fq;
        const double* sB = lds + (c & 1) * (BM + BN) * GS_LSTR + BM * GS_LSTR + (wn * WN * 16 + fr) * GS_LSTR + fq;
#pragma unroll
        for (int ks = 0; ks < GS_KC / 4; ++ks) {
            double af[WM], bf[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = sA[i * 16 * GS_LSTR + ks * 4];
#pragma unroll
            for (int j = 0; j < WN; ++j) bf[j] = sB[j * 16 * GS_LSTR + ks * 4];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (c + 1 < nk) swrite((c + 1) & 1);
        __syncthreads();
    }
    // accumulator map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = n0 + (wn * WN + j) * 16 + fr;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int row = m0 + (wm * WM + i) * 16 + fq + 4 * x;
                if (row < M && col < N) {
                    double* p = C + (int64_t)row * ldc + col;
                    double val = sign * acc[i][j][x];
                    if (beta) val += *p;
                    *p = val;
                }
            }
        }
}

// Read-out of the bordered factorisation: G = -(corner), sum of the per-block log-det partials.
// res[0..255] = G (16x16 row-major), res[256] = sum_i log L_ii, res[257] = info.
__global__ __launch_bounds__(256) void k_finalize(const double* A, int64_t ld, int np, const double* logdet,
                                                   int T, const int* info, double* res) {
    const int t = threadIdx.x;
    const int r = t >> 4, c = t & 15;
    res[t] = -A[(int64_t)(np + r) * ld + np + c];
    if (t == 0) {
        double s = 0.0;
        for (int i = 0; i < T; ++i) s += logdet[i];
        res[256] = s;
        res[257] = (double)(*info);
    }
}

// out[r] = sum_j B[r][j]^2 over ncols; one wave per row, fixed summation order.
__global__ __launch_bounds__(256) void k_rowsumsq(const double* B, int64_t ldb, int nrows, int ncols, double* out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const double* p = B + (int64_t)row * ldb;
    double s = 0.0;
    for (int j = lane; j < ncols; j += 64) s += p[j] * p[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) out[row] = s;
}

// Mirror the lower triangle into the upper one / zero the upper one, into a dense n x n buffer.
__global__ __launch_bounds__(256) void k_export(const double* A, int64_t ld, int n, double* out, int zero_upper) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n) return;
    double v;
    if (j <= i) v = A[(int64_t)i * ld + j];
    else v = zero_upper ? 0.0 : A[(int64_t)j * ld + i];
    out[(int64_t)i * n + j] = v;
}

// ---- probes ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_probe_mfma(double* out, int iters) {
    gs_d4 acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = (gs_d4){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[(int64_t)blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_probe_store(gs_d2* out, int64_t nvec) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    gs_d2 v = {1.0, 2.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) out[i] = v;
}
