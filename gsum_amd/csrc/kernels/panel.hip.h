// rows against a factored diagonal block: blocked substitution on the matrix cores, row IO, the sibling update
// (part of gsum_kernels.hip.h: included from there, in order; gfx950 only)
#pragma once
// ---- rows against a factored diagonal block: blocked substitution on the matrix cores ---------------------------
// X L_bb^T = B for 16 rows of B (128 columns), in place, by ONE wave, from the block's substitution tables in LDS: the
// rows are eight micro-blocks in the register image of gs_diag_block (P_k = -B_k^T), and column step j is
//     S   <- sum_{p<j} (-L_jp) P_p        (from zero, ascending p)
//     P_j <- D_j (P_j + S)                (X_j^T = D_j (B_j - sum_{p<j} X_p L_jp^T)^T)
// -- the arithmetic the rows below a diagonal micro-block go through inside gs_diag_block, with no pivoting work.  Only the
// 16 x 16 inverses D_j multiply, the off-diagonal part of L_bb enters through products with L itself: this is forward
// substitution at micro-block granularity, backward stable up to cond(L_jj) of 16 x 16 blocks, where a product with the
// explicit 128 x 128 inverse (round 1) loses cond(L_bb): measured against the extended-precision value of the S2 / S3
// log-likelihoods that product was 6-20 x further from the truth than LAPACK.  144 MFMAs per 16 rows instead of 256.
// rows: pointer to the first of the 16 rows at the block's first column; nvalid: rows that exist (others read as 0).
__device__ __forceinline__ void gs_panel16_load(gs_d4 (&P)[8], const double* rows, int64_t ld, int nvalid, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    const bool live = fr < nvalid;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int x = 0; x < 4; ++x) P[k][x] = live ? -rows[(int64_t)fr * ld + 16 * k + fq + 4 * x] : 0.0;
}

__device__ __forceinline__ void gs_panel16_solve(gs_d4 (&P)[8], const double* tab, int lane) {
    const double* Ls = tab + GS_D2_LS;
    const double* Dv = tab + GS_D2_DV;
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        gs_d4 S = {0.0, 0.0, 0.0, 0.0};                                // + sum_{p<j} L_jp X_p^T, from zero in ascending p
#pragma unroll
        for (int pp = 0; pp < j; ++pp) gs_d2_upd(S, Ls + (j * (j - 1) / 2 + pp) * 256, P[pp], lane);
        const gs_d4 E = P[j] + S;                                      // -(B_j - sum)^T: one subtraction of the whole sum
        gs_d4 T = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int x = 0; x < 4; ++x)
            T = __builtin_amdgcn_mfma_f64_16x16x4f64(Dv[(j * 16 + fr) * GS_DV_STR + fq + 4 * x], E[x], T, 0, 0, 0);
        P[j] = T;
    }
}

// two 16-row groups against the same tables in one pass: every table block is read from LDS once and feeds two
// independent MFMA chains (k_lml_medium's panel phase: 25-30 % of that kernel with one group at a time, each wave waiting
// on its own LDS reads and dependent MFMAs).  Row for row the arithmetic of gs_panel16_solve.
__device__ __forceinline__ void gs_panel16_solve2(gs_d4 (&P)[8], gs_d4 (&Q)[8], const double* tab, int lane) {
    const double* Ls = tab + GS_D2_LS;
    const double* Dv = tab + GS_D2_DV;
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        gs_d4 SP = {0.0, 0.0, 0.0, 0.0}, SQ = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int pp = 0; pp < j; ++pp) {
            const double* blk = Ls + (j * (j - 1) / 2 + pp) * 256;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const double a = blk[x * 64 + lane];
                SP = __builtin_amdgcn_mfma_f64_16x16x4f64(a, P[pp][x], SP, 0, 0, 0);
                SQ = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Q[pp][x], SQ, 0, 0, 0);
            }
        }
        const gs_d4 EP = P[j] + SP, EQ = Q[j] + SQ;
        gs_d4 TP = {0.0, 0.0, 0.0, 0.0}, TQ = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const double dv = Dv[(j * 16 + fr) * GS_DV_STR + fq + 4 * x];
            TP = __builtin_amdgcn_mfma_f64_16x16x4f64(dv, EP[x], TP, 0, 0, 0);
            TQ = __builtin_amdgcn_mfma_f64_16x16x4f64(dv, EQ[x], TQ, 0, 0, 0);
        }
        P[j] = TP;
        Q[j] = TQ;
    }
}

__device__ __forceinline__ void gs_panel16_store(const gs_d4 (&P)[8], double* rows, int64_t ld, int nvalid, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    if (fr < nvalid) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int x = 0; x < 4; ++x) rows[(int64_t)fr * ld + 16 * k + fq + 4 * x] = -P[k][x];
    }
}

// The same two through a wave-private LDS tile (16 rows x 18 doubles): the register image wants lane (fr, fq) to hold columns fq + 4 x of
// row fr, so gs_panel16_load's instruction x of micro-block k touches 16 rows x 32 B -- sixteen cache lines for 512 B, four times over
// per micro-block, and the stores are 32-B fragments.  Here a micro-block goes global <-> registers as TWO 16-B-per-lane accesses of 8
// whole 128-B lines each, and changes layout in LDS (a wave's LDS operations execute in order: no barrier).  Measured on the batch's
// panel launches (probe builds, profiles/r04_panel_rows.log): the fragmented row traffic was 4 ms of a 61-ms call.  Same values.
#define GS_PT_STR 18
#define GS_PT_TILE (16 * GS_PT_STR)
__device__ __forceinline__ void gs_panel16_load_t(gs_d4 (&P)[8], const double* rows, int64_t ld, int nvalid, int lane, double* tile) {
    const int fr = lane & 15, fq = lane >> 4;
    const int r0 = lane >> 3, cp = 2 * (lane & 7);
#pragma unroll
    for (int k = 0; k < 8; ++k)                  // raw lines into the registers the image will occupy: all 16 loads in flight together
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const gs_d2 v = (r0 + 8 * h < nvalid) ? *reinterpret_cast<const gs_d2*>(rows + (int64_t)(r0 + 8 * h) * ld + 16 * k + cp) : gs_d2{0.0, 0.0};
            P[k][2 * h] = v[0];
            P[k][2 * h + 1] = v[1];
        }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int h = 0; h < 2; ++h) *reinterpret_cast<gs_d2*>(tile + (r0 + 8 * h) * GS_PT_STR + cp) = gs_d2{P[k][2 * h], P[k][2 * h + 1]};
        gs_wave_lds_sync();
#pragma unroll
        for (int x = 0; x < 4; ++x) P[k][x] = -tile[fr * GS_PT_STR + fq + 4 * x];
        gs_wave_lds_sync();
    }
}

__device__ __forceinline__ void gs_panel16_store_t(const gs_d4 (&P)[8], double* rows, int64_t ld, int nvalid, int lane, double* tile) {
    const int fr = lane & 15, fq = lane >> 4;
    const int r0 = lane >> 3, cp = 2 * (lane & 7);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int x = 0; x < 4; ++x) tile[fr * GS_PT_STR + fq + 4 * x] = -P[k][x];
        gs_wave_lds_sync();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const gs_d2 v = *reinterpret_cast<const gs_d2*>(tile + (r0 + 8 * h) * GS_PT_STR + cp);
            if (r0 + 8 * h < nvalid) *reinterpret_cast<gs_d2*>(rows + (int64_t)(r0 + 8 * h) * ld + 16 * k + cp) = v;
        }
        gs_wave_lds_sync();
    }
}

__device__ __forceinline__ void gs_panel16(double* rows, int64_t ld, int nvalid, const double* tab, int lane) {
    gs_d4 P[8];
    gs_panel16_load(P, rows, ld, nvalid, lane);
    gs_panel16_solve(P, tab, lane);
    gs_panel16_store(P, rows, ld, nvalid, lane);
}

// global -> LDS copy of one block's substitution tables (256 threads); ends with a barrier
__device__ __forceinline__ void gs_load_ltab(double* tab, const double* Ltab) {
    const gs_d2* src = reinterpret_cast<const gs_d2*>(Ltab);
    gs_d2* dst = reinterpret_cast<gs_d2*>(tab);
    for (int i = threadIdx.x; i < GS_LTAB / 2; i += 256) dst[i] = src[i];
    __syncthreads();
}

// the same copy without staging registers: global_load_lds_dwordx4 moves each wave's 64 x 16 B straight into LDS
// (wave w of the workgroup's 4 takes every fourth 1-KiB piece); the caller waits (vmcnt) and synchronises
__device__ __forceinline__ void gs_load_ltab_direct(double* tab, const double* Ltab, int w, int lane) {
    constexpr int PIECES = GS_LTAB * 8 / 1024;                 // 73 whole 1-KiB pieces (GS_LTAB * 8 = 74752 = 73 KiB)
    for (int pc = w; pc < PIECES; pc += 4)
        __builtin_amdgcn_global_load_lds(Ltab + pc * 128 + 2 * lane, tab + pc * 128, 16, 0, 0);
}

// ---- the same substitution with the tables read straight from GLOBAL memory (L2 / L1 hits: every wave of a launch
// reads the same 73 KB), software-pipelined through registers: no LDS, no barrier, one wave per workgroup.  What it
// buys is placement, not arithmetic: beside the bulk update every CU holds three bulk workgroups and 1 KB of free LDS,
// and a 73-KB table workgroup waited for two of them to retire on the SAME CU (rocprofv3: 100-200 us per call in the
// first third of a factorisation, 20 us alone).  A lone wave with ~200 VGPRs and no LDS fits on any SIMD at once.
// Step j's table blocks (j panel dumps + D_j) are fetched one to two steps ahead; bit-identical to gs_panel16_solve.
template <int J>
__device__ __forceinline__ void gs_ptab_fetch(gs_d4 (&buf)[8], const double* tab, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int pp = 0; pp < J; ++pp)
#pragma unroll
        for (int x = 0; x < 4; ++x) buf[pp][x] = tab[GS_D2_LS + ((J * (J - 1) / 2 + pp) * 4 + x) * 64 + lane];
#pragma unroll
    for (int x = 0; x < 4; ++x) buf[J][x] = tab[GS_D2_DV + (J * 16 + fr) * GS_DV_STR + fq + 4 * x];
}

template <int J>
__device__ __forceinline__ void gs_ptab_step(gs_d4 (&P)[8], const gs_d4 (&buf)[8]) {
    gs_d4 S = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int pp = 0; pp < J; ++pp)
#pragma unroll
        for (int x = 0; x < 4; ++x) S = __builtin_amdgcn_mfma_f64_16x16x4f64(buf[pp][x], P[pp][x], S, 0, 0, 0);
    const gs_d4 E = P[J] + S;
    gs_d4 T = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int x = 0; x < 4; ++x) T = __builtin_amdgcn_mfma_f64_16x16x4f64(buf[J][x], E[x], T, 0, 0, 0);
    P[J] = T;
}

__device__ __forceinline__ void gs_panel16_solve_g(gs_d4 (&P)[8], const double* tab, int lane) {
    gs_d4 b0[8], b1[8], b2[8], b3[8];
    gs_ptab_fetch<0>(b0, tab, lane);
    gs_ptab_fetch<1>(b1, tab, lane);
    gs_ptab_fetch<2>(b2, tab, lane);
    gs_ptab_fetch<3>(b3, tab, lane);
    __builtin_amdgcn_sched_barrier(0);
    gs_ptab_step<0>(P, b0);
    gs_ptab_step<1>(P, b1);
    gs_ptab_fetch<4>(b0, tab, lane);
    __builtin_amdgcn_sched_barrier(0);
    gs_ptab_step<2>(P, b2);
    gs_ptab_step<3>(P, b3);
    gs_ptab_fetch<5>(b1, tab, lane);
    __builtin_amdgcn_sched_barrier(0);
    gs_ptab_step<4>(P, b0);
    gs_ptab_fetch<6>(b2, tab, lane);
    __builtin_amdgcn_sched_barrier(0);
    gs_ptab_step<5>(P, b1);
    gs_ptab_fetch<7>(b3, tab, lane);
    __builtin_amdgcn_sched_barrier(0);
    gs_ptab_step<6>(P, b2);
    gs_ptab_step<7>(P, b3);
}

// rows [0, M) x 128 columns at P (leading dimension ld)  <-  rows * L_bb^-T, 16 rows per single-wave workgroup
__global__ __launch_bounds__(64) void k_panel(double* P, int64_t ld, int M, const double* Ltab) {
    const int lane = threadIdx.x;
    const int r0 = blockIdx.x * 16;
    if (r0 >= M) return;
    __builtin_amdgcn_s_setprio(3);          // chain kernel: ahead of the bulk waves it shares the SIMD with
    double* rows = P + (int64_t)r0 * ld;
    gs_d4 Pr[8];
    gs_panel16_load(Pr, rows, ld, M - r0, lane);
    gs_panel16_solve_g(Pr, Ltab, lane);
    gs_panel16_store(Pr, rows, ld, M - r0, lane);
}

// ---- two block columns at once ------------------------------------------------------------------------------------
// Lsib: the 128 x 128 block L(j+1, j) as 64 micro-block dumps in A-operand layout, [(c * 8 + k) * 256 + x * 64 + lane]
// = register image of (-L_ck) -- what k_potrf_diag256 leaves behind for the rows below.
#define GS_LSIB (64 * 256)

// P1 (register image of the rows' second 128 columns)  +=  sum_k (-L_ck) P0_k : the sibling-column update
// B[:, j+1] -= X_j L(j+1, j)^T of these 16 rows, products in ascending k on accumulators that START as the matrix entries
// -- element for element the arithmetic of k_gemm_nt on the same block (sign-mirrored), so the fused kernels below stay
// bit-identical to the three-launch sequence panel / sibling update / panel.
__device__ __forceinline__ void gs_sib_fetch(gs_d4 (&buf)[8], const double* Lsib, int c, int lane) {
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int x = 0; x < 4; ++x) buf[kb][x] = Lsib[((c * 8 + kb) * 4 + x) * 64 + lane];
}

__device__ __forceinline__ void gs_sib_apply(gs_d4& acc, const gs_d4 (&buf)[8], const gs_d4 (&P0)[8]) {
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int x = 0; x < 4; ++x) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(buf[kb][x], P0[kb][x], acc, 0, 0, 0);
}

__device__ __forceinline__ void gs_sib_update(gs_d4 (&P1)[8], const gs_d4 (&P0)[8], const double* Lsib, int lane) {
    gs_d4 ba[8], bb[8];
    gs_sib_fetch(ba, Lsib, 0, lane);
#pragma unroll
    for (int c = 0; c < 8; c += 2) {
        gs_sib_fetch(bb, Lsib, c + 1, lane);
        __builtin_amdgcn_sched_barrier(0);
        gs_sib_apply(P1[c], ba, P0);
        if (c + 2 < 8) gs_sib_fetch(ba, Lsib, c + 2, lane);
        __builtin_amdgcn_sched_barrier(0);
        gs_sib_apply(P1[c + 1], bb, P0);
    }
}

// the same with half-size buffers (k-blocks 0..3 / 4..7 of one micro-block row at a time): 64 registers of operands in
// flight instead of 128.  For k_panel256 in a batch: six bulk waves (72 registers each) leave 80 of a SIMD's 512 registers
// free and every retiring bulk workgroup 144 more, so a wave of up to 224 registers starts where ONE bulk workgroup has
// left; a bigger one needs two or three gone and keeps them away for as long as it waits for memory.
template <int H>
__device__ __forceinline__ void gs_sib_fetch_half(gs_d4 (&buf)[4], const double* Lsib, int c, int lane) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int x = 0; x < 4; ++x) buf[kb][x] = Lsib[((c * 8 + 4 * H + kb) * 4 + x) * 64 + lane];
}

template <int H>
__device__ __forceinline__ void gs_sib_apply_half(gs_d4& acc, const gs_d4 (&buf)[4], const gs_d4 (&P0)[8]) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int x = 0; x < 4; ++x) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(buf[kb][x], P0[4 * H + kb][x], acc, 0, 0, 0);
}

__device__ __forceinline__ void gs_sib_update_lean(gs_d4 (&P1)[8], const gs_d4 (&P0)[8], const double* Lsib, int lane) {
    gs_d4 b0[4], b1[4];
    gs_sib_fetch_half<0>(b0, Lsib, 0, lane);
    gs_sib_fetch_half<1>(b1, Lsib, 0, lane);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        __builtin_amdgcn_sched_barrier(0);
        gs_sib_apply_half<0>(P1[c], b0, P0);                    // ascending k: blocks 0..3, then 4..7
        if (c + 1 < 8) gs_sib_fetch_half<0>(b0, Lsib, c + 1, lane);
        __builtin_amdgcn_sched_barrier(0);
        gs_sib_apply_half<1>(P1[c], b1, P0);
        if (c + 1 < 8) gs_sib_fetch_half<1>(b1, Lsib, c + 1, lane);
    }
}

