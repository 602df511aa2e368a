// back-substitution, predictive pieces, series scaling, sampling
// (part of gsum_kernels.hip.h: included from there, in order; gfx950 only)
#pragma once
// ---- back-substitution half of cho_solve (models.py:479): x^T L = w^T on the 16 border rows, right-looking from the
// last block column to the first.  Rows = right-hand sides (16), columns = points; everything on the matrix cores
// straight from global memory (the factor is read exactly once: HBM-bound, 4 n^2 bytes).
// One 16 x 128 panel times a 128 x 128 row-major matrix:  out[a][j] = sum_k P[a][k] M[k][j].  P comes from LDS
// (16 rows, stride 129), M from global memory (leading dimension ldm); wave w owns columns [32 w, 32 w + 32).
__device__ __forceinline__ void gs_panel_times_block(const double* P, const double* M, int64_t ldm, gs_d4 (&o)[2], int w, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    o[0] = (gs_d4){0.0, 0.0, 0.0, 0.0};
    o[1] = (gs_d4){0.0, 0.0, 0.0, 0.0};
    for (int s4 = 0; s4 < 32; ++s4) {
        const int kk = 4 * s4 + fq;
        const double av = P[fr * 129 + kk];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const double bv = M[(int64_t)kk * ldm + (2 * w + h) * 16 + fr];
            o[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, o[h], 0, 0, 0);
        }
    }
}

// X_c^T = W_c^T L_cc^-1 for the LAST block column (c0 = its first column), in place on the border rows.
__global__ __launch_bounds__(256) void k_back_first(double* Brow, int64_t ld, const double* Linv, int c0) {
    __shared__ double P[16 * 129];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int idx = t; idx < 16 * 128; idx += 256) P[(idx >> 7) * 129 + (idx & 127)] = Brow[(int64_t)(idx >> 7) * ld + c0 + (idx & 127)];
    __syncthreads();
    gs_d4 o[2];
    gs_panel_times_block(P, Linv, 128, o, w, lane);
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int x = 0; x < 4; ++x) Brow[(int64_t)(fq + 4 * x) * ld + c0 + (2 * w + h) * 16 + fr] = o[h][x];
}

// Step c (block column c holds the finished X_c^T): workgroup g < c subtracts X_c^T L[c rows, g cols] from W_g^T; the
// workgroup of block column c - 1 then finishes it, X_{c-1}^T = W_{c-1}^T L_{c-1,c-1}^-1 (nothing else touches it later).
__global__ __launch_bounds__(256) void k_back_step(const double* A, int64_t ld, double* Brow, const double* Linv, int c) {
    __shared__ double P[16 * 129];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = blockIdx.x, c0 = c * 128, g0 = g * 128;
    const int fr = lane & 15, fq = lane >> 4;
    for (int idx = t; idx < 16 * 128; idx += 256) P[(idx >> 7) * 129 + (idx & 127)] = -Brow[(int64_t)(idx >> 7) * ld + c0 + (idx & 127)];
    gs_d4 acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int x = 0; x < 4; ++x) acc[h][x] = Brow[(int64_t)(fq + 4 * x) * ld + g0 + (2 * w + h) * 16 + fr];
    __syncthreads();
    const double* Lblk = A + (int64_t)c0 * ld + g0;                 // rows of block c, columns of block g
    for (int s4 = 0; s4 < 32; ++s4) {
        const int kk = 4 * s4 + fq;
        const double av = P[fr * 129 + kk];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const double bv = Lblk[(int64_t)kk * ld + (2 * w + h) * 16 + fr];
            acc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[h], 0, 0, 0);
        }
    }
    if (g != c - 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int x = 0; x < 4; ++x) Brow[(int64_t)(fq + 4 * x) * ld + g0 + (2 * w + h) * 16 + fr] = acc[h][x];
        return;
    }
    __syncthreads();                                                // every wave is done reading P
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int x = 0; x < 4; ++x) P[(fq + 4 * x) * 129 + (2 * w + h) * 16 + fr] = acc[h][x];
    __syncthreads();
    gs_d4 o[2];
    gs_panel_times_block(P, Linv + (size_t)g * 128 * 128, 128, o, w, lane);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int x = 0; x < 4; ++x) Brow[(int64_t)(fq + 4 * x) * ld + g0 + (2 * w + h) * 16 + fr] = o[h][x];
}

// upper triangle <- lower triangle, in place (rows are written coalesced; the strided reads hit L2 for the sizes this serves)
__global__ __launch_bounds__(256) void k_mirror_lower(double* A, int64_t ld, int n) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j < n && j > i) A[(int64_t)i * ld + j] = A[(int64_t)j * ld + i];
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

// Coefficient covariance -> partial-sum covariance, in place (models.py:1343-1354 with helpers.py:149-182):
//   A_ij *= factor * ref_r[i] ref_c[j] * S(ratio_r[i] ratio_c[j]),
//   S(x) = (x^start - x^(end+1)) / (1 - x) - sum_{e excluded, start <= e <= end} x^e;   end < 0: infinite sum, x^(end+1) = 0.
// Same operation order as the reference's array expression; pow() is within an ulp of numpy's.
__global__ __launch_bounds__(256) void k_scale_series(double* A, int64_t ld, int rows, int cols, const double* ref_r,
                                                       const double* ratio_r, const double* ref_c, const double* ratio_c,
                                                       gsum_series_scale sc) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= cols || i >= rows) return;
    const double x = ratio_r[i] * ratio_c[j];
    const double hi = sc.end < 0 ? 0.0 : pow(x, (double)(sc.end + 1));
    double sum = (pow(x, (double)sc.start) - hi) / (1.0 - x);
    for (int e = 0; e < sc.n_excluded; ++e) {
        const int ex = sc.excluded[e];
        if (ex >= sc.start && (sc.end < 0 || ex <= sc.end)) sum -= pow(x, (double)ex);
    }
    const double refm = ref_r[i] * ref_c[j];
    A[(int64_t)i * ld + j] = (refm * sum) * (sc.factor * A[(int64_t)i * ld + j]);
}

// out = L Z for the lower-triangular factor (row-major, leading dimension ld), Z and out n x 16 (zero-padded
// columns): the sampling transform y = mean + L z of a multivariate normal.  One wave per row: lanes stride over the
// row's columns j <= i (coalesced 8-B loads of L, which is read exactly once: HBM-bound, n^2/2 x 8 B), 16
// accumulators per lane, then a butterfly reduction.  Z (n x 128 B) stays in L2.
__global__ __launch_bounds__(256) void k_tri_multiply(const double* L, int64_t ld, int n, const double* Z, double* out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    double acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.0;
    const double* row = L + (int64_t)i * ld;
    for (int j = lane; j <= i; j += 64) {
        const double l = row[j];
        const gs_d2* z = reinterpret_cast<const gs_d2*>(Z + (int64_t)j * 16);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const gs_d2 v = z[q];
            acc[2 * q] = __builtin_fma(l, v[0], acc[2 * q]);
            acc[2 * q + 1] = __builtin_fma(l, v[1], acc[2 * q + 1]);
        }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        double v = acc[c];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        acc[c] = v;
    }
    if (lane < 16) {
        double v = acc[0];
#pragma unroll
        for (int c = 1; c < 16; ++c) v = (lane == c) ? acc[c] : v;
        out[(int64_t)i * 16 + lane] = v;
    }
}

// Vt[c][j] = sum_{k >= j} U[j][k] Wt[c][k] for an upper-triangular U (row-major, n x n) and 16 rows Wt: V^T = W^T U^T of
// the gradient path.  One wave per row of U (read once, coalesced: HBM-bound, 4 n^2 bytes), 16 accumulators per lane,
// butterfly reduction.  (As a 16 x n x n GEMM on 16 x 256 tiles this had 32 workgroups with K = n each: 3.7 ms at n = 8192.)
__global__ __launch_bounds__(256) void k_upper_times_rows(const double* U, int64_t ldu, int n, const double* Wt, int64_t ldw,
                                                           double* Vt, int64_t ldv) {
    // FOUR rows of U per wave (round 4): a lane's 16 loads of W^T per k then feed 64 FMAs instead of 16 -- with one row per wave the kernel
    // was bound by those (L1-served) loads, 1.7 ms at n = 8192 for 268 MB of U.  Per row the same terms in the same order as before.
    const int lane = threadIdx.x & 63;
    const int j0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (j0 >= n) return;
    double acc[4][16];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[r][c] = 0.0;
    const double* row = U + (int64_t)j0 * ldu;
    for (int k = (j0 & ~63) + lane; k < n; k += 64) {
        double u[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = (j0 + r < n && k >= j0 + r) ? row[(int64_t)r * ldu + k] : 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const double w = Wt[(int64_t)c * ldw + k];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r][c] = __builtin_fma(u[r], w, acc[r][c]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            double v = acc[r][c];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            acc[r][c] = v;
        }
        if (lane < 16 && j0 + r < n) {
            double v = acc[r][0];
#pragma unroll
            for (int c = 1; c < 16; ++c) v = (lane == c) ? acc[r][c] : v;
            Vt[(int64_t)lane * ldv + j0 + r] = v;
        }
    }
}

