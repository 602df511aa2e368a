// whole evaluations in one workgroup: k_lml_small (n <= 128), k_lml_medium (128 < n <= 4096)
// (part of gsum_kernels.hip.h: included from there, in order; gfx950 only)
#pragma once
// ------------------------------------------------------------------------------------------------
// Fused path for n <= 128 (the reference's own problem sizes: 5-100 points, thousands of grid points):
// ONE workgroup per evaluation builds K, factors it, solves for the right-hand sides and reduces the Gram
// matrix; a launch evaluates a whole row of a likelihood grid.  Same arithmetic as the general path
// (k_build's kernel functions, gs_diag_block), per-evaluation scratch in global memory (L2-resident).
//   scratch per evaluation: A (128x128) | W^T (16x128, in a 128x128 slot);   res per evaluation: 258 doubles as k_finalize.
// ------------------------------------------------------------------------------------------------
#define GS_SMALL_SCRATCH (2 * 128 * 128)

// TREE: the instantiation the host launches when a call carries a Sum / Product tree (n_ops > 0): such an evaluation builds its matrix
// with gs_build_tile128_tree, every other evaluation of the call with the flattened code; a call of flattened descriptors only runs the
// TREE = false kernel, whose register budget the postfix walk never touches.
template <bool TREE>
__global__ __launch_bounds__(256, 2) void k_lml_small(const double* X, int n, int d, const double* Z, int k,
                                                    const gsum_kernel_desc* __restrict__ descs, double nugget, double* scratch,
                                                    double* res, const int32_t* zset) {
#pragma clang fp contract(off)
    __shared__ double dg0[128];
    __shared__ double ldet;
    __shared__ __attribute__((aligned(16))) double wsd[GS_DIAG_WS];     // lent to the build (us) and the solve (Wt) too:
    double* us = wsd;                                                   // 78.6 KB of LDS in all, two evaluations per CU
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const gsum_kernel_desc& desc = descs[blockIdx.x];      // in place (uniform address, read-only: scalar loads) -- a private copy is 712 B of scratch per lane
    double* A = scratch + (int64_t)blockIdx.x * GS_SMALL_SCRATCH;
    double* Wt = A + 128 * 128;                                         // W^T, 16 x 128 row-major (L2-resident)
    double* out = res + (int64_t)blockIdx.x * 258;
    if (zset) Z += (int64_t)zset[blockIdx.x] * n * k;                  // this evaluation's right-hand-side set (gsum_lml_resident_sets)
    // ---- kernel matrix (full symmetric 128x128 tile, identity padding beyond n)
    double* etab = us + 128 * GSUM_MAX_D;                               // exp tables th[16] | tl[16]
    if (t < 16) etab[t] = gs_exp_th[t];
    else if (t < 32) etab[t] = gs_exp_tl[t - 16];
    for (int idx = t; idx < 128 * d; idx += 256) {
        const int r = idx / d, dd = idx - r * d;
        const double ls = desc.anisotropic ? desc.length_scale[dd] : desc.length_scale[0];
        us[idx] = r < n ? X[(int64_t)r * d + dd] / ls : 0.0;
    }
    __syncthreads();
    // (the one tile is a diagonal tile: rows and columns are the same points; family / dimension as template parameters)
    if (TREE && descs[blockIdx.x].n_ops > 0) gs_build_tile128_tree(A, 128, X, 0, 0, n, d, descs[blockIdx.x], nugget, dg0, w, lane);
    else gs_build_tile128_any(A, 128, us, us, etab, etab + 16, 0, 0, n, d, desc, nugget, dg0, w, lane);
    __threadfence_block();
    __syncthreads();
    // ---- Cholesky of the block; its substitution tables stay in wsd
    // (only the micro-blocks that hold points are factorised: the rest of the block is identity padding, gs_d2_wave<.., PARTIAL>)
    const int bad = gs_diag_block<true, true>(A, 128, (double*)nullptr, (double*)nullptr, &ldet, dg0, nullptr, wsd, (n + 15) >> 4);
    if (bad) {
        if (t == 0) {
            out[256] = 0.0;
            out[257] = (double)bad;
        }
        return;
    }
    // ---- W^T = Z^T L^-T: the right-hand sides as 16 rows of 128 points, solved by one wave against the tables
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
    // ---- Gram matrix G = W^T W (16 x 16, K = 128) by wave 0
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
}

// ------------------------------------------------------------------------------------------------
// Fused path for 128 < n <= GS_MEDIUM_MAX (4096) when MANY evaluations are asked for (a likelihood grid on a few hundred
// to a couple of thousand points): ONE workgroup per evaluation runs the whole bordered pipeline on its own matrix
// in HBM (288 GB holds thousands of them), so a launch keeps 256 evaluations in flight, one per CU, with no
// inter-workgroup dependency and no per-step kernel launches.  Same building blocks as the general path: k_build's
// kernel functions, gs_diag_block, and a 128x128x(K = 128) MFMA tile routine with the accumulation order of
// k_gemm_nt (the trailing update is a plain right-looking sweep: per element it subtracts the same products in the
// same ascending-k order as the two-level schedule, so the factor is bit-identical to the general path's).
// Per-evaluation scratch: A (np x ld, ld = np + 16) | Linv (T x 128 x 128) | diag0 (np) | W^T (16 x np).
// ------------------------------------------------------------------------------------------------
#define GS_MEDIUM_MAX 4096
__device__ int gs_medium_lazy = 64;                 // option "medium_lazy": depth of the grouping of the fused sweep's trailing updates (1: one K = 256 update of every
                                                    // trailing tile per outer step; 2: K = 512 every other step; >= 16: LEFT-LOOKING at n <= 4096 -- a tile is read and
                                                    // written once, when its panel is next, with all the panels before it in one pass: the default, this sweep is HBM-bound)

// C (M x N, both <= 128) = beta C + sign A B^T with A: M x K, B: N x K, K a multiple of 16; 256 threads (2 x 2 waves of
// 64 x 64).  Operand chunks go global -> LDS directly (global_load_lds_dwordx4) in k_gemm_ld3's layout: XOR-swizzled
// k-pairs, even / odd rows in regions one double apart (no bank conflicts), the sign carried by negated accumulators.
// LDS: 2 stages x 2 operands x (128 x 16 + 2) doubles.  Ends with a workgroup barrier after the stores (fenced).
#define GS_TILE_LD_DOUBLES (2 * 2 * (128 * GS_KC + 2))
__device__ __forceinline__ void gs_tile128(double* C, int64_t ldc, const double* A, int64_t lda, const double* B, int64_t ldb,
                                           int M, int N, int K, int beta, double sign, double* lds) {
    constexpr int WM = 4, WN = 4;
    constexpr int OPER = 128 * GS_KC + 2, STAGE = 2 * OPER, HALF = 64 * GS_KC + 1;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w & 1, wn = w >> 1;
    const int fr = lane & 15, fq = lane >> 4;
    const bool neg = sign < 0.0;
    gs_d4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = (wn * WN + j) * 16 + fr;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int row = (wm * WM + i) * 16 + fq + 4 * x;
                const double c = (beta && row < M && col < N) ? C[(int64_t)row * ldc + col] : 0.0;
                acc[i][j][x] = neg ? -c : c;
            }
        }
    // this wave stages tile rows [32 w, 32 w + 32) of both operands: per parity h two loads of 8 rows each
    const int lrow = lane >> 3, lg = lane & 7;
    const double* srcA[2][2];
    const double* srcB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = 32 * w + 16 * q + 2 * lrow + h;
            const int kp = lg ^ ((r >> 1) & 7);
            const int ra = r < M ? r : M - 1, rb = r < N ? r : N - 1;
            srcA[h][q] = A + (int64_t)ra * lda + 2 * kp;
            srcB[h][q] = B + (int64_t)rb * ldb + 2 * kp;
        }
    auto stage_load = [&](int kc, int stage) {
        double* base = lds + stage * STAGE;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                double* dstA = base + h * HALF + (16 * w + 8 * q) * GS_KC;
                double* dstB = base + OPER + h * HALF + (16 * w + 8 * q) * GS_KC;
                __builtin_amdgcn_global_load_lds(srcA[h][q] + kc * GS_KC, dstA, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(srcB[h][q] + kc * GS_KC, dstB, 16, 0, 0);
            }
    };
    const int swz = (fr >> 1) & 7;
    const int rsel = (fr & 1) * HALF + (fr >> 1) * GS_KC;
    int goff[GS_KC / 4];
#pragma unroll
    for (int ks = 0; ks < GS_KC / 4; ++ks) goff[ks] = (((2 * ks + (fq >> 1)) ^ swz) << 1) + (fq & 1);
    const int nk = K / GS_KC;
    __syncthreads();                                     // the previous user of `lds` is done with it
    stage_load(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = 0; c < nk; ++c) {
        if (c + 1 < nk) stage_load(c + 1, (c + 1) & 1);
        const double* sA = lds + (c & 1) * STAGE + wm * WM * 8 * GS_KC + rsel;
        const double* sB = lds + (c & 1) * STAGE + OPER + wn * WN * 8 * GS_KC + rsel;
#pragma unroll
        for (int ks = 0; ks < GS_KC / 4; ++ks) {
            double af[WM], bf[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = sA[i * 8 * GS_KC + goff[ks]];
#pragma unroll
            for (int j = 0; j < WN; ++j) bf[j] = sB[j * 8 * GS_KC + goff[ks]];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = (wn * WN + j) * 16 + fr;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int row = (wm * WM + i) * 16 + fq + 4 * x;
                if (row < M && col < N) C[(int64_t)row * ldc + col] = neg ? -acc[i][j][x] : acc[i][j][x];
            }
        }
    __threadfence_block();
    __syncthreads();
}

template <bool TREE>
__global__ __launch_bounds__(256, 2) void k_lml_medium(const double* X, int n, int d, const double* Z, int k,
                                                     const gsum_kernel_desc* __restrict__ descs, double nugget, double* scratch,
                                                     int64_t scratch_stride, double* res, unsigned long long* stamps, const int32_t* zset) {
    extern __shared__ double lds[];                 // max(GS_DIAG_WS, GS_TILE_LD_DOUBLES) doubles, lent in turn to the kernel
                                                    // build, the diagonal-block routine and the tile routine: 77.6 KB in
                                                    // all, so TWO evaluations share a CU
    __shared__ double ldet_blk;
    __shared__ double ldet_sum;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const gsum_kernel_desc& desc = descs[blockIdx.x];      // in place (uniform address, read-only: scalar loads) -- a private copy is 712 B of scratch per lane
    const int np = (n + 127) / 128 * 128, T = np / 128;
    const int64_t ld = GS_LD(np);
    double* A = scratch + (int64_t)blockIdx.x * scratch_stride;
    double* diag0 = A + (int64_t)np * ld + (int64_t)T * 128 * 128;     // (the T x 128 x 128 slot before it held exported tables
                                                                       // while the right-hand sides had a sweep of their own)
    double* Wt = diag0 + np;                        // 16 x np, row-major
    double* out = res + (int64_t)blockIdx.x * 258;
    if (zset) Z += (int64_t)zset[blockIdx.x] * n * k;                  // this evaluation's right-hand-side set (gsum_lml_resident_sets)
    // diagnostics (option "diag_stamps"): shader cycles of workgroup 0 per phase -> stamps[40..47] =
    // {build, diagonal blocks, panel solves, sibling tiles, trailing tiles, W step, Gram + rest, total}
    unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0}, tq = 0, tstart = 0;
    const bool stamping = stamps != nullptr && blockIdx.x == 0 && t == 0;
    auto phase = [&](int i) {
        if (stamping) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (i >= 0) ph[i] += now - tq; else tstart = now;
            tq = now;
        }
    };
    phase(-1);
    // ---- kernel matrix: lower 128x128 tiles, identity padding (k_build's arithmetic)
    {
#pragma clang fp contract(off)
        double* ui = lds;
        double* uj = lds + 128 * GSUM_MAX_D;
        double* etab = lds + 2 * 128 * GSUM_MAX_D;      // exp tables th[16] | tl[16] (first read behind the loop's barriers)
        if (t < 16) etab[t] = gs_exp_th[t];
        else if (t < 32) etab[t] = gs_exp_tl[t - 16];
        for (int bi = 0; bi < T; ++bi)
            for (int bj = 0; bj <= bi; ++bj) {
                __syncthreads();
                for (int idx = t; idx < 128 * d; idx += 256) {
                    const int r = idx / d, dd = idx - r * d;
                    const double ls = desc.anisotropic ? desc.length_scale[dd] : desc.length_scale[0];
                    const int gi = bi * 128 + r, gj = bj * 128 + r;
                    ui[idx] = gi < n ? X[(int64_t)gi * d + dd] / ls : 0.0;
                    uj[idx] = gj < n ? X[(int64_t)gj * d + dd] / ls : 0.0;
                }
                __syncthreads();
                if (TREE && descs[blockIdx.x].n_ops > 0) gs_build_tile128_tree(A, ld, X, bi, bj, n, d, descs[blockIdx.x], nugget, diag0, w, lane);
                else gs_build_tile128_any(A, ld, ui, uj, etab, etab + 16, bi, bj, n, d, desc, nugget, diag0, w, lane);
            }
    }
    if (t == 0) ldet_sum = 0.0;
    __threadfence_block();
    __syncthreads();
    phase(0);
    const int fr = lane & 15, fq = lane >> 4;
    // ---- right-looking blocked Cholesky, two block columns per trailing update (K = 256: the trailing tiles are read
    // and written once per 256 eliminated columns, which is what this HBM-resident sweep is bound by)
    int grp = 0;                                    // outer steps of the current group already applied to the NEXT panel's columns only
    for (int b = 0; b < T; b += 2) {
        const bool two = b + 1 < T;
        for (int s = 0; s < (two ? 2 : 1); ++s) {
            const int c = b + s;
            // (no table export: every consumer of block c's tables -- the panel below, right-hand-side rows included -- reads
            // them from LDS before the next block overwrites them)
            const int bad = gs_diag_block(A + (int64_t)c * 128 * ld + c * 128, ld, (double*)nullptr, (double*)nullptr,
                                          &ldet_blk, diag0 + c * 128, nullptr, lds);
            if (bad) {
                if (t == 0) {
                    out[256] = 0.0;
                    out[257] = (double)(c * 128 + bad);
                }
                return;
            }
            if (t == 0) ldet_sum += ldet_blk;
            __threadfence_block();
            __syncthreads();
            phase(1);
            // The 16 right-hand-side rows are rows of the bordered matrix: block column c of W^T = Z^T L^-T is brought up to
            // date here (left-looking over the columns already done, on the matrix cores straight from global memory: wave w
            // owns point-columns [32 w, 32 w + 32) of the block) and then SOLVED WITH THE PANEL below, against the tables
            // gs_diag_block has just left in LDS.  As a separate sweep after the factorisation every block cost a reload of its
            // 73-KB table, two barriers and a lone wave solving while three waited: 12-15 % of the kernel at n <= 1024.
            {
                gs_d4 acc[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int pc = c * 128 + (2 * w + h) * 16 + fr;        // accumulator column = point index
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const int rr = fq + 4 * x;                         // accumulator row = right-hand side
                        acc[h][x] = (rr < k && pc < n) ? Z[(int64_t)pc * k + rr] : 0.0;
                    }
                }
                // minus W^T[:, c'] L[c, c']^T, ascending k; eight k-steps requested at a time before their MFMAs
                const double* wrow = Wt + (int64_t)fr * np + fq;
                const double* l0 = A + (int64_t)(c * 128 + (2 * w) * 16 + fr) * ld + fq;
                const double* l1 = l0 + (int64_t)16 * ld;
                for (int kk0 = 0; kk0 < c * 128; kk0 += 32) {
                    double av[8], bv0[8], bv1[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        av[u] = -wrow[kk0 + 4 * u];
                        bv0[u] = l0[kk0 + 4 * u];
                        bv1[u] = l1[kk0 + 4 * u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv0[u], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv1[u], acc[1], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int x = 0; x < 4; ++x) Wt[(int64_t)(fq + 4 * x) * np + c * 128 + (2 * w + h) * 16 + fr] = acc[h][x];
            }
            __threadfence_block();
            __syncthreads();
            phase(5);
            // panel: rows below <- rows * L_cc^-T by blocked substitution against the tables in lds, two 16-row groups per
            // wave and pass; the last group is the right-hand-side rows
            {
                const int ngr = (T - c - 1) * 8;        // 16-row groups of the matrix below the block; group ngr = W^T[:, c]
                double* pan = A + ((int64_t)(c + 1) * 128) * ld + c * 128;
                for (int br = w; br <= ngr; br += 8) {
                    double* ra = br < ngr ? pan + (int64_t)(16 * br) * ld : Wt + c * 128;
                    const int64_t lda_ = br < ngr ? ld : np;
                    if (br + 4 <= ngr) {
                        double* rb = br + 4 < ngr ? pan + (int64_t)(16 * (br + 4)) * ld : Wt + c * 128;
                        const int64_t ldb_ = br + 4 < ngr ? ld : np;
                        gs_d4 Pg[8], Qg[8];
                        gs_panel16_load(Pg, ra, lda_, 16, lane);
                        gs_panel16_load(Qg, rb, ldb_, 16, lane);
                        gs_panel16_solve2(Pg, Qg, lds, lane);
                        gs_panel16_store(Pg, ra, lda_, 16, lane);
                        gs_panel16_store(Qg, rb, ldb_, 16, lane);
                    } else {
                        gs_panel16(ra, lda_, 16, lds, lane);
                    }
                }
            }
            __threadfence_block();
            if (stamping || stamps) __syncthreads();        // (diagnostic runs only: a barrier so that the phases separate)
            phase(2);
            if (s == 0 && two)                      // sibling block column b + 1: the first panel only (K = 128)
                for (int i = b + 1; i < T; ++i)
                    gs_tile128(A + (int64_t)i * 128 * ld + (b + 1) * 128, ld, A + (int64_t)i * 128 * ld + b * 128, ld,
                               A + (int64_t)(b + 1) * 128 * ld + b * 128, ld, 128, 128, 128, 1, -1.0, lds);
            phase(3);
        }
        const int Kp = two ? 256 : 128, first = b + (two ? 2 : 1);
        // The batch factorisation's pairing of trailing updates (lazy_far = 2) inside this sweep: after an even outer step only the NEXT two block columns take
        // this panel's update (K = 256); the step after it applies both panels to every tile right of them in one K = 512 pass -- the trailing tiles, which this
        // HBM-resident sweep reads and writes once per update, are then touched half as often.  Same products in the same ascending-k order per element.
        // gs_medium_lazy = depth of the grouping (1: none, 2: pairs, d: the far tiles are touched once per d outer steps, with K = 256 d)
        const int depth = gs_medium_lazy;
        const bool more = depth > 1 && two && first + 1 < T && grp + 1 < depth;      // a full two-block panel follows and the group is not complete
        const int gb = b - 2 * grp;                                                  // first block column of the group: [gb, b + 2) are (grp + 1) x 256 contiguous columns
        const int Kg = 256 * grp + Kp;
        const int jlast = more ? first + 1 : T - 1;                                  // near update: the next panel's two block columns only
        for (int i = first; i < T; ++i)
            for (int j = first; j <= min(i, jlast); ++j)
                gs_tile128(A + (int64_t)i * 128 * ld + j * 128, ld, A + (int64_t)i * 128 * ld + gb * 128, ld,
                           A + (int64_t)j * 128 * ld + gb * 128, ld, 128, 128, Kg, 1, -1.0, lds);
        grp = more ? grp + 1 : 0;
        phase(4);
    }
    __threadfence_block();
    __syncthreads();                               // W^T complete (its last block was solved by whichever wave had the group)
    // ---- Gram matrix G = W^T W by wave 0 (ascending k), log-det, info
    if (w == 0) {
        gs_d4 g = {0.0, 0.0, 0.0, 0.0};
        for (int s4 = 0; s4 < np / 4; ++s4) {
            const double wv = Wt[(int64_t)fr * np + 4 * s4 + fq];
            g = __builtin_amdgcn_mfma_f64_16x16x4f64(wv, wv, g, 0, 0, 0);
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) out[(fq + 4 * x) * 16 + fr] = g[x];
        if (lane == 0) {
            out[256] = ldet_sum;
            out[257] = 0.0;
        }
    }
    phase(6);
    if (stamping) {
        for (int i = 0; i < 7; ++i) stamps[40 + i] = ph[i];
        stamps[47] = tq - tstart;
    }
}

