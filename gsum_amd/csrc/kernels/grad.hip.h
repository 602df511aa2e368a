// gradient path: kernel-gradient contractions and their reductions
// (part of gsum_kernels.hip.h: included from there, in order; gfx950 only)
#pragma once
// ---- gradient path (models.py:957-958, 1041-1056) ------------------------------------------------
__global__ __launch_bounds__(256) void k_set_identity(double* A, int64_t ld, int np) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j < np) A[(int64_t)i * ld + j] = (i == j) ? 1.0 : 0.0;
}

struct gs_grad_params { gsum_grad_param p[GSUM_MAX_GRAD]; };

// d kernel(X)_ij / d theta_p for one log-hyperparameter (scikit-learn's K_gradient; kernels.py of sklearn 1.x:
// RBF.__call__, Matern.__call__, ConstantKernel, WhiteKernel, Product / Sum rules), evaluated on the fly.
// s = sum_m D_m, D_m = ((x_im - x_jm) / l_m)^2;  dm = s for an isotropic length scale, D_dim otherwise.
__device__ __forceinline__ double gs_kernel_grad(const gsum_kernel_desc& desc, const gsum_grad_param& pr, double s, double dm,
                                                 bool diag) {
    switch (pr.code) {
        case GSUM_GRAD_AMPLITUDE: return desc.amplitude * (diag ? 1.0 : gs_base_value(desc.family, s));
        case GSUM_GRAD_WHITE: return diag ? pr.weight : 0.0;
        case GSUM_GRAD_ADDITIVE: return pr.weight;
        default: break;
    }
    if (diag) return 0.0;
    double g;
    if (desc.family == GSUM_RBF) {
        g = gs_base_value(GSUM_RBF, s) * dm;
    } else if (desc.family == GSUM_MATERN52) {
        const double tmp = sqrt(5.0 * s);
        g = 5.0 / 3.0 * dm * (tmp + 1.0) * gs_exp_np(-tmp);
    } else if (desc.family == GSUM_MATERN32) {
        g = 3.0 * dm * gs_exp_np(-sqrt(3.0 * s));
    } else {
        const double den = sqrt(s);
        g = den != 0.0 ? gs_base_value(GSUM_MATERN12, s) * (dm / den) : 0.0;
    }
    return desc.amplitude * g;
}

// One wave per row i of dR_p (grid.y = p): Q_p[i][c] = sum_j dR_p,ij V[j][c] (V^T given as 16 rows: coalesced loads)
// and trow_p[i] = sum_{j<=i} (2 - [i == j]) Rinv_ij dR_p,ij, so that sum_i trow = tr(R^-1 dR_p) from the lower
// triangle of R^-1 alone.  dR is never stored: n^2 kernel-gradient evaluations per parameter, HBM traffic = the
// lower triangle of R^-1 once per parameter.
// TREE: the descriptor is a Sum / Product tree (n_ops > 0).  Two instantiations: with the tree walk in the same kernel the flattened
// form -- every kernel the reference itself constructs -- ran at 218 registers instead of 146 (two waves per SIMD instead of three) and the
// whole gradient evaluation 5 % slower.
// R rows of dR_p per wave (round 4): the 16 loads of V^T (and X_j) per column j then feed R x 16 FMAs -- with one row per wave those
// L1-served loads bound the kernel (1.3 ms per parameter at n = 8192).  Per row the same terms in the same order as before.
// SPLIT (round 5, one evaluation alone): Q_p only, and dR_p,ij for j <= i goes to dR (P lower triangles, leading dimension ldr) -- this half needs
// V^T alone and runs on the vector pipes BESIDE the R^-1 = U U^T product on the matrix pipes; k_grad_trace then takes the traces from R^-1 and the
// stored triangle with the same terms in the same order per lane (bit-identical to the fused form, which a batch keeps: its other members fill the chip).
template <bool TREE, int R, bool SPLIT = false>
__global__ __launch_bounds__(256) void k_grad_contract(const double* X, int n, int d, gsum_kernel_desc desc, gs_grad_params prm,
                                                        const double* Rinv, int64_t ldr, const double* Vt, int64_t ldv,
                                                        double* Q, double* trow, double* dR = nullptr, int64_t dr_stride = 0) {
    const int lane = threadIdx.x & 63;
    const int i0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    const int p = blockIdx.y;
    if (i0 >= n) return;
    const gsum_grad_param pr = prm.p[p];
    double xi[R][GSUM_MAX_D], inv_ls[GSUM_MAX_D];
#pragma unroll
    for (int m = 0; m < GSUM_MAX_D; ++m) {
        inv_ls[m] = 1.0 / (desc.anisotropic ? desc.length_scale[m < d ? m : 0] : desc.length_scale[0]);
#pragma unroll
        for (int r = 0; r < R; ++r) xi[r][m] = (m < d && i0 + r < n) ? X[(int64_t)(i0 + r) * d + m] : 0.0;
    }
    double acc[R][16], tr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        tr[r] = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[r][c] = 0.0;
    }
    for (int j = lane; j < n; j += 64) {
        double xj[GSUM_MAX_D];
#pragma unroll
        for (int m = 0; m < GSUM_MAX_D; ++m) xj[m] = m < d ? X[(int64_t)j * d + m] : 0.0;
        double g[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = i0 + r;
            double s = 0.0, dsel = 0.0;
#pragma unroll
            for (int m = 0; m < GSUM_MAX_D; ++m) {
                if (m < d) {
                    const double u = (xi[r][m] - xj[m]) * inv_ls[m];
                    const double dmm = u * u;
                    s += dmm;
                    if (m == pr.dim) dsel = dmm;
                }
            }
            const double dm = pr.code == GSUM_GRAD_LENGTH_ISO ? s : dsel;
            if constexpr (TREE) (void)gs_tree_eval(desc, xi[r], xj, d, i == j, &pr, &g[r]);      // a general tree: the kernel build's walk, with dual numbers
            else g[r] = gs_kernel_grad(desc, pr, s, dm, i == j);
            if constexpr (SPLIT) {
                if (i < n && j <= i) dR[(int64_t)p * dr_stride + (int64_t)i * ldr + j] = g[r];
            } else {
                if (i < n && j <= i) tr[r] = __builtin_fma((j < i ? 2.0 : 1.0) * Rinv[(int64_t)i * ldr + j], g[r], tr[r]);
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const double v = Vt[(int64_t)c * ldv + j];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][c] = __builtin_fma(g[r], v, acc[r][c]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (i0 + r >= n) break;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            double v = acc[r][c];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            acc[r][c] = v;
        }
        double t = tr[r];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
        if (lane < 16) {
            double v = acc[r][0];
#pragma unroll
            for (int c = 1; c < 16; ++c) v = (lane == c) ? acc[r][c] : v;
            Q[((int64_t)p * n + i0 + r) * 16 + lane] = v;
        }
        if (!SPLIT && lane == 0) trow[(int64_t)p * n + i0 + r] = t;
    }
}

// the other half of the split form: trow_p[i] = sum_{j <= i} (2 - [i == j]) Rinv_ij dR_p,ij from the stored triangle, one wave per row, lane l
// taking j = l, l + 64, ... in ascending order and the same xor tree over the lanes as k_grad_contract: the same bits
__global__ __launch_bounds__(256) void k_grad_trace(const double* Rinv, int64_t ldr, const double* dR, int64_t dr_stride, int n, double* trow) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int p = blockIdx.y;
    if (i >= n) return;
    const double* rr = Rinv + (int64_t)i * ldr;
    const double* dd = dR + (int64_t)p * dr_stride + (int64_t)i * ldr;
    double t = 0.0;
    for (int j = lane; j <= i; j += 64) t = __builtin_fma((j < i ? 2.0 : 1.0) * rr[j], dd[j], t);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
    if (lane == 0) trow[(int64_t)p * n + i] = t;
}

// H_p = V^T Q_p (16 x 16) and sum_i trow_p[i], in two deterministic stages.  Stage 1 (grid: chunks x P): chunk c
// reduces rows [c * rows_per, (c + 1) * rows_per) into part[(p * chunks + c) * 257 ...]; stage 2 (grid: P) adds the
// chunks in index order.
__global__ __launch_bounds__(256) void k_grad_reduce1(const double* Vt, int64_t ldv, const double* Q, const double* trow, int n,
                                                       int rows_per, double* part) {
    __shared__ double red[256];
    const int t = threadIdx.x, a = t >> 4, b = t & 15, c = blockIdx.x, p = blockIdx.y;
    const int lo = c * rows_per, hi = min(n, lo + rows_per);
    const double* Qp = Q + (int64_t)p * n * 16;
    double h = 0.0;
    for (int i = lo; i < hi; ++i) h = __builtin_fma(Vt[(int64_t)a * ldv + i], Qp[(int64_t)i * 16 + b], h);
    double* o = part + ((int64_t)p * gridDim.x + c) * 257;
    o[t] = h;
    double ts = 0.0;
    for (int i = lo + t; i < hi; i += 256) ts += trow[(int64_t)p * n + i];
    red[t] = ts;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
    }
    if (t == 0) o[256] = red[0];
}

__global__ __launch_bounds__(256) void k_grad_reduce2(const double* part, int chunks, double* out) {
    const int t = threadIdx.x, p = blockIdx.x;
    const double* src = part + (int64_t)p * chunks * 257;
    double h = 0.0;
    for (int c = 0; c < chunks; ++c) h += src[(int64_t)c * 257 + t];
    out[(int64_t)p * 257 + t] = h;
    if (t == 0) {
        double ts = 0.0;
        for (int c = 0; c < chunks; ++c) ts += src[(int64_t)c * 257 + 256];
        out[(int64_t)p * 257 + 256] = ts;
    }
}

