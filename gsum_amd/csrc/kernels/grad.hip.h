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


// ------------------------------------------------------------------------------------------------
// Value + gradient pieces for n <= 128 in ONE workgroup per evaluation (round 5): the reference's own problem sizes are 5-100 points
// and its fit() drives L-BFGS with log_marginal_likelihood(theta, eval_gradient=True) (models.py:634-640, 957-958, 1041-1056) -- the
// general path's dozen launches and two synchronisations cost 250 us per objective evaluation there, this kernel ~60.
//   k_lml_small's steps (same code: G, sum log diag and info equal the value path's bit for bit), the explicit block inverse from
//   gs_diag_block, R^-1 = L^-T L^-1 and V^T = W^T L^-1 on the matrix cores (gs_tile128), then per hyperparameter dR_p entry by entry
//   (k_grad_contract's formulas), trace_p = tr(R^-1 dR_p) on the way, Q_p = dR_p V on the matrix cores, H_p = V^T Q_p.
//   scratch per evaluation: A | W^T (16 rows) and V^T (16 rows) in one slot | L^-1 | L^-T | R^-1  (five 128 x 128 slots);
//   res: 258 doubles as k_finalize;  gres: P x 257 (H_p 16 x 16, then the trace).
// ------------------------------------------------------------------------------------------------
#define GS_GSMALL_SCRATCH (5 * 128 * 128)
template <bool TREE>
__global__ __launch_bounds__(256, 2) void k_grad_small(const double* X, int n, int d, const double* Z, int k,
                                                     const gsum_kernel_desc* __restrict__ descs, const gsum_grad_param* __restrict__ params,
                                                     int P, double nugget, double* scratch, double* res, double* gres) {
#pragma clang fp contract(off)
    __shared__ double dg0[128];
    __shared__ double ldet;
    __shared__ __attribute__((aligned(16))) double wsd[GS_DIAG_WS];     // the build's points, the block's tables, the tile's stages, the contractions' rows
    static_assert(GS_DIAG_WS >= GS_TILE_LD_DOUBLES, "one LDS workspace serves every phase");
    double* us = wsd;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const gsum_kernel_desc& desc = descs[blockIdx.x];
    double* A = scratch + (int64_t)blockIdx.x * GS_GSMALL_SCRATCH;
    double* Wt = A + 128 * 128;                                         // W^T: 16 x 128
    double* Vt = Wt + 16 * 128;                                         // V^T: 16 x 128
    double* Linv = A + 2 * 128 * 128;
    double* Ut = A + 3 * 128 * 128;                                     // L^-T, row-major
    double* Rinv = A + 4 * 128 * 128;
    double* out = res + (int64_t)blockIdx.x * 258;
    double* gout = gres + (int64_t)blockIdx.x * P * 257;
    // ---- kernel matrix, factorisation, W^T, Gram matrix: k_lml_small, statement for statement
    double* etab = us + 128 * GSUM_MAX_D;
    if (t < 16) etab[t] = gs_exp_th[t];
    else if (t < 32) etab[t] = gs_exp_tl[t - 16];
    for (int idx = t; idx < 128 * d; idx += 256) {
        const int r = idx / d, dd = idx - r * d;
        const double ls = desc.anisotropic ? desc.length_scale[dd] : desc.length_scale[0];
        us[idx] = r < n ? X[(int64_t)r * d + dd] / ls : 0.0;
    }
    __syncthreads();
    if (TREE && descs[blockIdx.x].n_ops > 0) gs_build_tile128_tree(A, 128, X, 0, 0, n, d, descs[blockIdx.x], nugget, dg0, w, lane);
    else gs_build_tile128_any(A, 128, us, us, etab, etab + 16, 0, 0, n, d, desc, nugget, dg0, w, lane);
    __threadfence_block();
    __syncthreads();
    const int bad = gs_diag_block<true, true>(A, 128, Linv, (double*)nullptr, &ldet, dg0, nullptr, wsd, (n + 15) >> 4);
    if (bad) {
        if (t == 0) {
            out[256] = 0.0;
            out[257] = (double)bad;
        }
        return;
    }
    for (int idx = t; idx < 16 * 128; idx += 256) {
        const int c = idx >> 7, i = idx & 127;
        Wt[idx] = (c < k && i < n) ? Z[(int64_t)i * k + c] : 0.0;
    }
    __threadfence_block();
    __syncthreads();
    if (w == 0) gs_panel16(Wt, 128, 16, wsd, lane);
    __threadfence_block();
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    if (w == 0) {
        gs_d4 g = {0.0, 0.0, 0.0, 0.0};
        for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const double wv = Wt[fr * 128 + 16 * kb + 4 * s4 + fq];
                g = __builtin_amdgcn_mfma_f64_16x16x4f64(wv, wv, g, 0, 0, 0);
            }
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) out[(fq + 4 * x) * 16 + fr] = g[x];
        if (lane == 0) {
            out[256] = ldet;
            out[257] = 0.0;
        }
    }
    // ---- L^-T (the transpose of the block inverse), then on the matrix cores V^T = W^T L^-1 = W^T (L^-T)^T and R^-1 = L^-T (L^-T)^T
    // (beyond n the factor is the identity: everything below works on the leading n16 x n16 part, n16 = n rounded up to the tiles' K step)
    const int n16 = (n + 15) & ~15;
    {
        const int c = t & 127, r0 = t >> 7;                             // thread t writes column c of rows r0, r0 + 2, ...: reads along a row of L^-1
        if (c < n16) {
#pragma unroll 8
            for (int r = r0; r < n16; r += 2) Ut[r * 128 + c] = Linv[c * 128 + r];
        }
    }
    __threadfence_block();
    __syncthreads();                                                   // (the tables in wsd are no longer needed: the tile's stages take their place)
    // (rows 16 .. 127 of the A operand are whatever the slot holds: they feed accumulators that are never stored)
    gs_tile128(Vt, 128, Wt, 128, Ut, 128, 16, n16, n16, 0, 1.0, wsd);
    gs_tile128(Rinv, 128, Ut, 128, Ut, 128, n16, n16, n16, 0, 1.0, wsd);
    // ---- contractions, as matrix products (round 5, second form: one wave per two rows walked the rows too slowly beyond n = 64):
    //   dR_p (n16 x n16, kernel-gradient entries on the fly, 256 threads abreast) -> the slot A held;  trace_p = sum_ij R^-1_ij dR_p,ij on the way;
    //   Q_p = dR_p V on the matrix cores (gs_tile128: A = dR_p, B = V^T);  H_p = V^T Q_p from LDS copies.
    double* xs = wsd + GS_TILE_LD_DOUBLES;                             // 128 x GSUM_MAX_D, beside the tile's stages
    double* red = xs + 128 * GSUM_MAX_D;                               // 256
    static_assert(GS_DIAG_WS >= GS_TILE_LD_DOUBLES + 128 * GSUM_MAX_D + 256, "the contractions' LDS beside the tile's stages");
    static_assert(GS_TILE_LD_DOUBLES >= 2 * 16 * 128, "Q_p and V^T fit where the tile's stages were");
    double* dRm = A;                                                   // (L is no longer needed: the block inverse and W^T carry everything)
    double* Qg = Ut;                                                   // n16 x 16 (L^-T is no longer needed either)
    for (int idx = t; idx < 128 * GSUM_MAX_D; idx += 256) {
        const int r = idx / GSUM_MAX_D, m = idx - r * GSUM_MAX_D;
        xs[idx] = (r < n && m < d) ? X[(int64_t)r * d + m] : 0.0;
    }
    double inv_ls[GSUM_MAX_D];
#pragma unroll
    for (int m = 0; m < GSUM_MAX_D; ++m) inv_ls[m] = 1.0 / (desc.anisotropic ? desc.length_scale[m < d ? m : 0] : desc.length_scale[0]);
    __syncthreads();
    for (int p = 0; p < P; ++p) {
        const gsum_grad_param pr = params[(int64_t)blockIdx.x * P + p];
        double tr = 0.0;
        for (int idx = t; idx < n16 * n16; idx += 256) {
            const int i = idx / n16, j = idx - i * n16;
            double g = 0.0;
            if (i < n && j < n) {
                bool walked = false;
                if constexpr (TREE) {
                    if (desc.n_ops > 0) {
                        (void)gs_tree_eval(desc, xs + i * GSUM_MAX_D, xs + j * GSUM_MAX_D, d, i == j, &pr, &g);
                        walked = true;
                    }
                }
                if (!walked) {
                    double s2 = 0.0, dsel = 0.0;
#pragma unroll
                    for (int m = 0; m < GSUM_MAX_D; ++m)
                        if (m < d) {
                            const double u = (xs[i * GSUM_MAX_D + m] - xs[j * GSUM_MAX_D + m]) * inv_ls[m];
                            const double dmm = u * u;
                            s2 += dmm;
                            if (m == pr.dim) dsel = dmm;
                        }
                    g = gs_kernel_grad(desc, pr, s2, pr.code == GSUM_GRAD_LENGTH_ISO ? s2 : dsel, i == j);
                }
                tr = __builtin_fma(Rinv[i * 128 + j], g, tr);
            }
            dRm[i * 128 + j] = g;
        }
        red[t] = tr;
        __threadfence_block();
        __syncthreads();
        if (t == 0) {                                                  // the trace: 256 partial sums in index order (deterministic)
            double ts = 0.0;
            for (int q = 0; q < 256; ++q) ts += red[q];
            gout[(int64_t)p * 257 + 256] = ts;
        }
        gs_tile128(Qg, 16, dRm, 128, Vt, 128, n16, 16, n16, 0, 1.0, wsd);          // Q_p[i][c] = sum_j dR_p[i][j] V^T[c][j]
        {
            double* qs = wsd;                                          // Q_p: n16 x 16;  V^T: 16 x 128 behind it
            double* vs = wsd + 16 * 128;
            for (int idx = t; idx < n16 * 16; idx += 256) qs[idx] = Qg[idx];
            for (int idx = t; idx < 16 * 128; idx += 256) vs[idx] = (idx & 127) < n16 ? Vt[idx] : 0.0;
            __syncthreads();
            const int a2 = t >> 4, b2 = t & 15;                        // H_p = V^T Q_p, rows in index order
            double h = 0.0;
            for (int i = 0; i < n; ++i) h = __builtin_fma(vs[a2 * 128 + i], qs[i * 16 + b2], h);
            gout[(int64_t)p * 257 + t] = h;
        }
        __syncthreads();
    }
}
