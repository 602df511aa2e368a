// K2a: the 128 x 128 diagonal block in MFMA accumulator registers (gs_diag_block)
// (part of gsum_kernels.hip.h: included from there, in order; gfx950 only)
#pragma once
// ------------------------------------------------------------------------------------------------
// K2a: diagonal block (128x128), one workgroup.  (The round-1 routine -- the block in the registers of a 16 x 16 thread grid, an
// LDS mailbox round trip per two columns, the explicit 128 x 128 inverse -- was removed in round 4; gs_diag_block below is what runs.)
// LAPACK dpotf2 semantics: pivot <= 0 or NaN -> info, plus the guard of gs_pivot_guard (top of this file).
// ------------------------------------------------------------------------------------------------
#define GS_DV_STR 17     // padded row stride of the 16x16 diagonal inverses in LDS

__device__ __forceinline__ double gs_rsqrt_nr(double p) {
    // ~1 ulp reciprocal square root: hardware estimate + two Newton-Raphson steps
    double r = __builtin_amdgcn_rsq(p);
    const double h = 0.5 * p;
    double e = __builtin_fma(-(h * r), r, 0.5);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-(h * r), r, 0.5);
    r = __builtin_fma(r, e, r);
    return r;
}

#define GS_DIAG_WS 9792      // LDS doubles the fused small / medium kernels reserve for the diagonal routine (76.5 KB >= its 9472)

// ------------------------------------------------------------------------------------------------
// K2a, second version (round 2): the 128x128 diagonal block as an 8 x 8 grid of 16x16 micro-blocks that live in the
// ACCUMULATOR REGISTERS of the four waves for the whole factorisation; only the pivot recurrence of one 16x16 micro-block
// at a time is scalar work, and it runs inside ONE wave with no barrier and no LDS round trip per column.
//
// Register image.  Wave w owns block rows w and 7 - w (9 micro-blocks, 72 VGPRs).  For micro-block (i, k) with current
// value M the four registers hold  P[x](lane l) = -M[l & 15][(l >> 4) + 4 x].  Read as an MFMA accumulator that is
// -M^T; read as the A operand of k-step x it is -M; read as the B operand of k-step x it is -M^T (all three with the
// same k numbering kappa(x, g) = g + 4 x).  Hence, with no data movement and no negation anywhere:
//   panel solve   L_ij^T = D_j M_ij^T :  P_ij <- mfma(A = D_j from LDS,             B = P_ij)         (D_j = L_jj^-1)
//   update        M_ik^T -= L_kj L_ij^T:  P_ik <- mfma(A = -L_kj = P_kj dumped to LDS, B = P_ij, C = P_ik)
// and the LDS copy of the panel ("dump": register x of lane l at [x][l], conflict-free both ways) is also the A operand
// -L_ip the block inverse needs afterwards.
// Pivot recurrence (gs_potf2_16): the owner wave turns its micro-block into one row per lane (lanes 0..15) with the
// rows of the identity beside it (lanes 16..31).  Column step c: the pivot and the scaled column entries l_k come out
// of their lanes with v_readlane into SGPRs, every lane does a[k] -= a[c] l_k -- the same instruction stream gives L in
// lanes 0..15 and L^-T in lanes 16..31 (column operations applied to the identity), so the micro-block inverse D_j
// costs nothing.  ~45 instructions per column instead of a barrier + mailbox round trip (~1200 cycles) per two columns.
// Schedule per micro-block column j: [barrier] panel solve + dump [barrier] the owner of row j + 1 updates its diagonal
// micro-block and runs the pivot recurrence at once while the other waves apply the remaining updates.
// Semantics unchanged: LAPACK dpotf2's pivot test plus the lost-every-bit threshold (see above), the first failing column
// reported; L in place (lower part only), L^-1 of the whole block to Linv, sum of log L_jj.
// Workspace: 28 panel blocks (56 KB) + 8 micro-block inverses (17 KB) + 128 thresholds = 9472 doubles <= GS_DIAG_WS.
// ------------------------------------------------------------------------------------------------
#define GS_D2_LS 0
#define GS_D2_DV (28 * 256)
#define GS_D2_THR (GS_D2_DV + 8 * 16 * GS_DV_STR)

__device__ __forceinline__ double gs_readlane_f64(double v, int srclane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void gs_wave_lds_sync() {
    // LDS operations of one wave execute in order; this only keeps the compiler from moving accesses across the point
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Accumulation order (what the log-likelihood's last digits depend on; measured against the extended-precision values of
// tests/golden/large_truth.json).  An entry of the factor is (a_ik - sum_p l_ip l_kp) / l_kk.  Subtracting the products
// from a_ik one by one, as a right-looking update does, rounds every partial result at the magnitude of a_ik, although
// the products of far-away columns are tiny and only the last few are large: on the S2 / S3 inputs (pivots ~ 1e-8 of the
// diagonal) that put the log-likelihood 6-20 x further from the true value than LAPACK.  Here the products are summed
// FROM ZERO in ascending p, in accumulators of their own (S), and the sum is subtracted once -- the order of a
// left-looking dot product -- while the schedule stays right-looking.  The pivot recurrence continues the same sums.
// Pivot recurrence of micro-block JB in one wave, entirely in the register image -- no LDS round trip, no per-lane row
// arrays.  Pjj: the ORIGINAL block (-A_jj^T, symmetric), Sjj: the products accumulated so far (+sum_p L_jp L_jp^T).  Column
// c = g + 4 x of a symmetric 16 x 16 matrix M sits in register x of the sixteen lanes of group g (lane & 15 = row), so
//   e   = A[x] - S[x]                     column c of the Schur complement on group g (one subtraction of the sum)
//   p   = readlane(e, 16 g + c)           the pivot;  1 / sqrt(p) by v_rsq + two Newton steps, uniform
//   m   = e / sqrt(p)                     column c of L on group g;  ms = its part strictly below the diagonal, else 0
//   S  += ms ms^T                         ONE v_mfma_f64_16x16x4 with ms as A and B operand in k-slot g (the other three
//                                         slots are zeros): the rank-1 update of all 256 sums
//   V   : column c scaled, V -= v_c l^T   a second MFMA (A = -ms, B = v_c): the column operations applied to the identity,
//                                         V -> L^-T, whose transpose D_j = L_jj^-1 goes to the table row-major
// ~35 instructions per column, two of them MFMAs, against ~70 VALU + v_readlane for a row-per-lane formulation (9.3 k
// cycles per micro-block measured) and a barrier + LDS mailbox per two columns in round 1.
// Ablk: the block's origin in the matrix.  Dvj: its 16 x 17 slot of the inverse table.  Returns the failing local column or -1.
// PAIR IMAGE of the diagonal micro-blocks.  The recurrence below eliminates two columns per step (the two pivots'
// reciprocal square roots are independent dependent-chains; done one after the other they are most of a column's ~430
// cycles), and for that both columns of a pair must sit in the same lanes.  A diagonal micro-block M (symmetric) is
// therefore held permuted: with rho(i) = (i >> 2) + 4 (i & 3), register x of lane l holds M[rho^-1(l & 15)][4 (l >> 4) + x]
// -- columns 4 g .. 4 g + 3 in the sixteen lanes of group g.  Seen as an MFMA accumulator this is Pi M Pi^T for the
// permutation Pi of rho, so rank-1 updates with vectors indexed the same way (lane & 15 = rho(row)) need nothing else:
// the sums S_jj reach it by reading the panel dumps with a permuted lane index (gs_d2_upd_diag), A_jj by loading it so.
__device__ __forceinline__ int gs_pair_row(int lane) { return 4 * (lane & 3) + ((lane & 15) >> 2); }   // rho^-1(lane & 15)
#define GS_PAIR_LANE(r, c) (16 * ((c) >> 2) + ((r) >> 2) + 4 * ((r) & 3))                              // lane of entry (r, c)

// v is zero outside lane row g (16 lanes): the same values in row g ^ 1, zero elsewhere.  v_permlane16_swap exchanges the odd
// rows of its first operand with the even rows of its second; ODD = g & 1.
__device__ __forceinline__ double gs_row_to_sibling(double v, bool ODD) {     // ODD folds after unrolling
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    if (ODD) {
        const auto a = __builtin_amdgcn_permlane16_swap(lo, 0u, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, 0u, false, false);
        return __hiloint2double((int)b[1], (int)a[1]);
    } else {
        const auto a = __builtin_amdgcn_permlane16_swap(0u, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(0u, hi, false, false);
        return __hiloint2double((int)b[0], (int)a[0]);
    }
}

template <int JB>
__device__ __forceinline__ int gs_potf2_16(const gs_d4& Ajj, const gs_d4& Sjj, double* Ablk, int64_t ld, double* Dvj,
                                           const double* thr, double* dbuf, int lane, unsigned long long* stamps = nullptr) {
    const int fq = lane >> 4, rr = gs_pair_row(lane);
    unsigned long long tq0 = 0;
    if (stamps) tq0 = __builtin_amdgcn_s_memtime();
    gs_d4 Aa = Ajj, S = Sjj, V, Lo = {0.0, 0.0, 0.0, 0.0}, Rs = {1.0, 1.0, 1.0, 1.0};
#pragma unroll
    for (int x = 0; x < 4; ++x) V[x] = (rr == 4 * fq + x) ? 1.0 : 0.0;
    const double tl = thr[16 * JB + rr];
    int fail = -1;
#pragma unroll
    for (int c0 = 0; c0 < 16; c0 += 2) {
        const int c1 = c0 + 1, g = c0 >> 2, x0 = c0 & 3, x1 = x0 + 1;
        const bool ing = fq == g;
        const double e0 = Aa[x0] - S[x0];     // columns c0, c1 of the Schur complement before c0 is eliminated (group g):
        const double e1 = Aa[x1] - S[x1];     // ONE subtraction of the zero-start sums each
        const double p0 = gs_readlane_f64(e0, GS_PAIR_LANE(c0, c0));
        const double a10 = gs_readlane_f64(e0, GS_PAIR_LANE(c1, c0));
        const double p1r = gs_readlane_f64(e1, GS_PAIR_LANE(c1, c1));
        const double t0 = gs_readlane_f64(tl, GS_PAIR_LANE(c0, 0) & 15), t1 = gs_readlane_f64(tl, GS_PAIR_LANE(c1, 0) & 15);
        // q = p0 p1 with p1 = p1r - a10^2 / p0 the second pivot: 1 / sqrt(p1) = sqrt(p0) rsqrt(q), so rsqrt(q) runs beside
        // rsqrt(p0) instead of behind it
        const double q = __builtin_fma(p1r, p0, -(a10 * a10));
        if (fail < 0 && !(p0 > t0)) fail = c0;            // wave-uniform (the operands came through SGPRs); catches NaN
        if (fail < 0 && !(q > t1 * p0)) fail = c1;        // <=> p1 <= threshold
        const double r0 = gs_rsqrt_nr(p0), rq = gs_rsqrt_nr(q);
        double d0 = p0 * r0;                                             // sqrt(p0) ...
        d0 = __builtin_fma(__builtin_fma(-d0, d0, p0), 0.5 * r0, d0);    // ... corrected to ~0.5 ulp
        double sq = q * rq;
        sq = __builtin_fma(__builtin_fma(-sq, sq, q), 0.5 * rq, sq);
        const double l10 = a10 * r0, r1 = d0 * rq, d1 = sq * r0;        // L[c1][c0], 1 / sqrt(p1), sqrt(p1)
        const double m0 = e0 * r0;
        const double ms0 = (ing && rr > c0) ? m0 : 0.0;
        // column c1 after c0: its sum takes the product l_r,c0 l_c1,c0 first, then the one subtraction
        const double m1 = (Aa[x1] - __builtin_fma(m0, l10, S[x1])) * r1;
        const double ms1 = (ing && rr > c1) ? m1 : 0.0;
        Lo[x0] = ing ? ((rr == c0) ? d0 : ms0) : Lo[x0];
        Lo[x1] = ing ? ((rr == c1) ? d1 : ms1) : Lo[x1];
        Rs[x0] = ing ? r0 : Rs[x0];           // the columns of V are scaled at the end (never updated after their step)
        Rs[x1] = ing ? r1 : Rs[x1];
        const double vc0 = ing ? V[x0] * r0 : 0.0;
        const double vc1 = ing ? __builtin_fma(-l10, vc0, V[x1]) * r1 : 0.0;
        // both rank-1 updates of the pair in ONE MFMA each: column c1's vector moves to the sibling lane row (g ^ 1, a
        // different k-slot) with v_permlane16_swap, so the instruction sums ms0 ms0^T + ms1 ms1^T (two of its four k-slots)
        const double ab = ms0 + gs_row_to_sibling(ms1, (g & 1) != 0);
        const double vb = vc0 + gs_row_to_sibling(vc1, (g & 1) != 0);
        S = __builtin_amdgcn_mfma_f64_16x16x4f64(ab, ab, S, 0, 0, 0);
        V = __builtin_amdgcn_mfma_f64_16x16x4f64(-ab, vb, V, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);    // keep the steps apart: hoisting the next steps' lane masks and v_readlane
                                              // results ahead ran the kernel out of SGPRs (spills through v_writelane)
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int cc = 4 * fq + x;
        if (cc <= rr) Ablk[(int64_t)rr * ld + cc] = Lo[x];                    // L_jj, lower part
        if (cc == rr) dbuf[16 * JB + rr] = Lo[x];                             // its diagonal, for the log-determinant
        Dvj[cc * GS_DV_STR + rr] = V[x] * Rs[x];                              // D_j[a][b] = V[b][a], row-major
    }
    if (stamps && lane == 0) stamps[24 + JB] = __builtin_amdgcn_s_memtime() - tq0;      // diagnostics: cycles of this recurrence
    return fail;
}

// C (acc) += A (dumped block at `blk`: [x][lane]) * B (registers)
__device__ __forceinline__ void gs_d2_upd(gs_d4& Cc, const double* blk, const gs_d4& Bb, int lane) {
#pragma unroll
    for (int x = 0; x < 4; ++x) Cc = __builtin_amdgcn_mfma_f64_16x16x4f64(blk[x * 64 + lane], Bb[x], Cc, 0, 0, 0);
}

// the same for a DIAGONAL micro-block's sum, kept in the pair image: both operands are the dumped block read with the
// pair image's lane index (row rho^-1(lane & 15) of the block)
__device__ __forceinline__ void gs_d2_upd_diag(gs_d4& Cc, const double* blk, int lane) {
    const int src = gs_pair_row(lane) + 16 * (lane >> 4);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const double v = blk[x * 64 + src];
        Cc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, Cc, 0, 0, 0);
    }
}

// X^T = D_J (A^T - sum)  in the register image:  P <- mfma(D_J, P + S), dumped to the panel table
// FULL: all 28 dumps stay in LDS (the fused kernels solve rows against them afterwards).  !FULL: LDS holds only the
// CURRENT panel column (slot = block row; a column is read in its own step only) and every dump goes straight to the
// global table Lg -- 35 KB of LDS instead of 77, so the stand-alone kernel fits into the place ONE bulk workgroup
// leaves behind on a busy CU.
#define GS_LS_SLOT(FULL, row, J) ((FULL) ? ((row) * ((row) - 1) / 2 + (J)) : (row))
template <int J, bool FULL>
__device__ __forceinline__ void gs_d2_solve_dump(gs_d4& Pb, const gs_d4& Sb, const double (&av)[4], double* Ls, double* Lg, double* A,
                                                 int64_t ld, int row, int lane) {
    const gs_d4 E = Pb + Sb;
    gs_d4 T = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int x = 0; x < 4; ++x) T = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x], E[x], T, 0, 0, 0);
    Pb = T;
#pragma unroll
    for (int x = 0; x < 4; ++x) Ls[(GS_LS_SLOT(FULL, row, J) * 4 + x) * 64 + lane] = T[x];

    if constexpr (!FULL) {
#pragma unroll
        for (int x = 0; x < 4; ++x) Lg[((row * (row - 1) / 2 + J) * 4 + x) * 64 + lane] = T[x];
    }
}

template <int W, int J, bool FULL>
__device__ __forceinline__ void gs_d2_trsm_dump(gs_d4 (&P0)[W + 1], gs_d4 (&P1)[8 - W], gs_d4 (&S0)[W + 1], gs_d4 (&S1)[8 - W],
                                                const double* Dv, double* Ls, double* Lg, double* A, int64_t ld, int lane) {
    constexpr int R0 = W, R1 = 7 - W;
    const int fr = lane & 15, fq = lane >> 4;
    if constexpr (R1 > J) {                   // R1 >= R0: nothing to do for either row otherwise
        double av[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) av[x] = Dv[(J * 16 + fr) * GS_DV_STR + fq + 4 * x];
        if constexpr (R0 > J) gs_d2_solve_dump<J, FULL>(P0[J], S0[J], av, Ls, Lg, A, ld, R0, lane);
        gs_d2_solve_dump<J, FULL>(P1[J], S1[J], av, Ls, Lg, A, ld, R1, lane);
    }
}

// after panel column J is in LDS: add its products to the sums; the owner of row J + 1 finishes the sum of its diagonal
// micro-block first and runs the pivot recurrence on it before its other updates.  Returns the failing local column of
// micro-block J + 1 or -1.
// the blocks of panel column J this wave solved are final: back to the matrix.  Issued in the update phase -- by the
// wave that runs the next pivot recurrence only after it, by the others first -- so the scattered 8-byte stores are off
// the chain (at the end of the kernel they were 8 k cycles of tail, in the solve phase 1-1.5 k per step)
template <int W, int J>
__device__ __forceinline__ void gs_d2_store_col(const gs_d4 (&P0)[W + 1], const gs_d4 (&P1)[8 - W], double* A, int64_t ld, int lane) {
    constexpr int R0 = W, R1 = 7 - W;
    const int fr = lane & 15, fq = lane >> 4;
    if constexpr (R0 > J) {
#pragma unroll
        for (int x = 0; x < 4; ++x) A[(int64_t)(16 * R0 + fr) * ld + 16 * J + fq + 4 * x] = -P0[J][x];
    }
    if constexpr (R1 > J) {
#pragma unroll
        for (int x = 0; x < 4; ++x) A[(int64_t)(16 * R1 + fr) * ld + 16 * J + fq + 4 * x] = -P1[J][x];
    }
}

template <int W, int J, bool FULL>
__device__ __forceinline__ int gs_d2_update(gs_d4 (&P0)[W + 1], gs_d4 (&P1)[8 - W], gs_d4 (&S0)[W + 1], gs_d4 (&S1)[8 - W], double* A,
                                            int64_t ld, double* Dv, double* scr, const double* Ls, const double* thr, double* dbuf,
                                            int lane, unsigned long long* stamps) {
    constexpr int R0 = W, R1 = 7 - W, N = J + 1;
    int fail = -1;
    if constexpr (R0 != N && R1 != N) gs_d2_store_col<W, J>(P0, P1, A, ld, lane);
    if constexpr (R0 == N) {
        gs_d2_upd_diag(S0[N], Ls + GS_LS_SLOT(FULL, N, J) * 256, lane);
        fail = gs_potf2_16<N>(P0[N], S0[N], A + (int64_t)(16 * N) * ld + 16 * N, ld, Dv + N * 16 * GS_DV_STR, thr, dbuf, lane, stamps);
    } else if constexpr (R1 == N) {
        gs_d2_upd_diag(S1[N], Ls + GS_LS_SLOT(FULL, N, J) * 256, lane);
        fail = gs_potf2_16<N>(P1[N], S1[N], A + (int64_t)(16 * N) * ld + 16 * N, ld, Dv + N * 16 * GS_DV_STR, thr, dbuf, lane, stamps);
    }
    if constexpr (R0 == N || R1 == N) gs_d2_store_col<W, J>(P0, P1, A, ld, lane);
    if constexpr (R0 > N) {
#pragma unroll
        for (int k = N; k < R0; ++k) gs_d2_upd(S0[k], Ls + GS_LS_SLOT(FULL, k, J) * 256, P0[J], lane);
        gs_d2_upd_diag(S0[R0], Ls + GS_LS_SLOT(FULL, R0, J) * 256, lane);           // the row's own diagonal micro-block
    }
    if constexpr (R1 > N) {
#pragma unroll
        for (int k = N; k < R1; ++k) gs_d2_upd(S1[k], Ls + GS_LS_SLOT(FULL, k, J) * 256, P1[J], lane);
        gs_d2_upd_diag(S1[R1], Ls + GS_LS_SLOT(FULL, R1, J) * 256, lane);
    }
    return fail;
}

// the blocks of panel column K that wave W owns, from the matrix into the register image: a strictly lower block negated
// in the standard image, a diagonal micro-block in the pair image, not negated, read from its lower triangle only
template <int W, int K>
__device__ __forceinline__ void gs_d2_load_col(gs_d4 (&P0)[W + 1], gs_d4 (&P1)[8 - W], const double* A, int64_t ld, int lane) {
    constexpr int R0 = W, R1 = 7 - W;
    const int fr = lane & 15, fq = lane >> 4;
    const int prow = gs_pair_row(lane);
    if constexpr (K <= R0) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            if constexpr (K < R0) {
                P0[K][x] = -A[(int64_t)(16 * R0 + fr) * ld + 16 * K + fq + 4 * x];
            } else {
                const int cc = 4 * fq + x, hi = cc > prow ? cc : prow, lo = cc > prow ? prow : cc;
                P0[K][x] = A[(int64_t)(16 * R0 + hi) * ld + 16 * R0 + lo];
            }
        }
    }
    if constexpr (K <= R1) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            if constexpr (K < R1) {
                P1[K][x] = -A[(int64_t)(16 * R1 + fr) * ld + 16 * K + fq + 4 * x];
            } else {
                const int cc = 4 * fq + x, hi = cc > prow ? cc : prow, lo = cc > prow ? prow : cc;
                P1[K][x] = A[(int64_t)(16 * R1 + hi) * ld + 16 * R1 + lo];
            }
        }
    }
}

template <int W, int J, bool FULL>
__device__ __forceinline__ bool gs_d2_step(gs_d4 (&P0)[W + 1], gs_d4 (&P1)[8 - W], gs_d4 (&S0)[W + 1], gs_d4 (&S1)[8 - W], double* A,
                                           int64_t ld, double* Dv, double* scr, double* Ls, double* Lg, const double* thr, double* dbuf,
                                           int* fail_sh, int lane, unsigned long long* stamps) {
    if constexpr (J < 7) gs_d2_load_col<W, J + 1>(P0, P1, A, ld, lane);       // next column's blocks: a step ahead of their use
    __syncthreads();                                          // D_J (and a failure flag) visible
    if (stamps && W == 0 && lane == 0) stamps[8 + 2 * J] = __builtin_amdgcn_s_memtime();       // diagnostics only
    if (*fail_sh >= 0) return false;
    gs_d2_trsm_dump<W, J, FULL>(P0, P1, S0, S1, Dv, Ls, Lg, A, ld, lane);
    __syncthreads();                                          // panel column J visible
    if (stamps && W == 0 && lane == 0) stamps[9 + 2 * J] = __builtin_amdgcn_s_memtime();
    if constexpr (J < 7) {
        const int f = gs_d2_update<W, J, FULL>(P0, P1, S0, S1, A, ld, Dv, scr, Ls, thr, dbuf, lane, stamps);
        if (f >= 0 && lane == 0) *fail_sh = 16 * (J + 1) + f;
    }
    return true;
}

// phase 1 of wave W: returns false if a pivot failed (every wave leaves at the same barrier)
// PARTIAL (round 5; the one-block kernels k_lml_small / k_grad_small only, FULL tables): only the first `nbv` micro-blocks hold data, the rest of
// the block is identity padding -- the reference's own orders are 5-20 points, and the factorisation of a 128 x 128 block that is seven eighths
// identity cost the same 36 us.  Steps 0 .. nbv - 2 run as always (they factor micro-blocks 0 .. nbv - 1); what the skipped steps would leave
// behind is written directly: zero panel dumps from column nbv - 1 on (the rows below are padding), identity micro-inverses and unit diagonal
// entries from block nbv on.  The matrix itself already holds the identity there.  Every other caller instantiates PARTIAL = false: the same code as before.
template <int W, bool FULL, bool PARTIAL = false>
__device__ __forceinline__ bool gs_d2_wave(double* A, int64_t ld, double* Dv, double* scr, double* Ls, double* Lg, double* thr,
                                           double d0, double* dbuf, int* fail_sh, int lane, unsigned long long* stamps, int nbv = 8) {
    static_assert(!PARTIAL || FULL, "the partial form keeps its tables in LDS");
    constexpr int R0 = W, R1 = 7 - W;
    gs_d4 P0[R0 + 1], P1[R1 + 1], S0[R0 + 1], S1[R1 + 1];

#pragma unroll
    for (int k = 0; k <= R0; ++k) S0[k] = (gs_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k <= R1; ++k) S1[k] = (gs_d4){0.0, 0.0, 0.0, 0.0};
    // Only panel column 0 is fetched here; column k + 1 is requested at the start of step k (gs_d2_step), a whole
    // step (7-10 k cycles) ahead of its use.  A block of the matrix is needed exactly once -- when its column is solved
    // (or, for a diagonal micro-block, factored) -- so holding all nine of a wave's blocks from the start only cost
    // registers: 72 of them, which is what pushed the fused kernels (capped at 256 for two evaluations per CU) into
    // scratch.  Nothing writes a block before it is read: stores go to columns already solved.
    gs_d2_load_col<W, 0>(P0, P1, A, ld, lane);
    // pivot thresholds (d0 was requested before the block, so it is the oldest load in flight): waves 0 and 1 store 64
    // each.  No barrier: the first recurrence reads entries 0..15, which its own wave wrote (LDS operations of one wave
    // execute in order); every later reader is behind the barriers of step 0.
    if constexpr (W < 2) thr[threadIdx.x] = d0 > 0.0 ? d0 * gs_pivot_guard : 0.0;
    if constexpr (W == 0) {
        gs_wave_lds_sync();
        const int f = gs_potf2_16<0>(P0[0], S0[0], A, ld, Dv, thr, dbuf, lane, stamps);
        if (f >= 0 && lane == 0) *fail_sh = f;
    }
#define GS_D2_STEP(J)                                                                                                             \
    if (!PARTIAL || nbv >= 8 || (J) + 1 < nbv) {              /* (a full block runs every step: step 7's barriers publish block 7) */ \
        if (!gs_d2_step<W, J, FULL>(P0, P1, S0, S1, A, ld, Dv, scr, Ls, Lg, thr, dbuf, fail_sh, lane, stamps)) return false;      \
    }
    GS_D2_STEP(0)
    GS_D2_STEP(1)
    GS_D2_STEP(2)
    GS_D2_STEP(3)
    GS_D2_STEP(4)
    GS_D2_STEP(5)
    GS_D2_STEP(6)
    GS_D2_STEP(7)
#undef GS_D2_STEP
    // (the strictly lower micro-blocks went back to the matrix as they were solved, the diagonal ones from the pivot recurrence)
    if constexpr (PARTIAL) {
        if (nbv < 8) {
            __syncthreads();                                  // the last recurrence's D, diagonal entries and failure flag are visible
            if (*fail_sh >= 0) return false;
            // what steps nbv - 1 .. 7 would have left: wave W writes for its own two block rows, like the steps do
            constexpr int R0 = W, R1 = 7 - W;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int row = half ? R1 : R0;
                for (int J = (nbv > 1 ? nbv - 1 : 0); J < row; ++J)
#pragma unroll
                    for (int x = 0; x < 4; ++x) Ls[(GS_LS_SLOT(true, row, J) * 4 + x) * 64 + lane] = 0.0;
                if (row >= nbv) {
                    for (int e = lane; e < 16 * GS_DV_STR; e += 64) Dv[row * 16 * GS_DV_STR + e] = (e / GS_DV_STR == e % GS_DV_STR) ? 1.0 : 0.0;
                    if (lane < 16) dbuf[16 * row + lane] = 1.0;
                }
            }
            __syncthreads();                                  // (a full step ends behind a barrier too: the tables and diagonal entries are visible)
        }
    }
    return true;
}

// One block column J of L^-1 on the matrix cores, from the panel dumps (A operand: -L_ip) and the micro-block inverses.
template <int J>
__device__ __forceinline__ void gs_trtri_col2(const double* Ls, const double* Dv, double* Linv, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    gs_d4 X[8];
#pragma unroll
    for (int x = 0; x < 4; ++x) X[J][x] = Dv[(J * 16 + fq + 4 * x) * GS_DV_STR + fr];      // X_JJ = D_J in accumulator layout
#pragma unroll
    for (int i = J + 1; i < 8; ++i) {
        gs_d4 T = {0.0, 0.0, 0.0, 0.0};                                                      // -sum_p L_ip X_pJ
#pragma unroll
        for (int p = J; p < i; ++p) gs_d2_upd(T, Ls + (i * (i - 1) / 2 + p) * 256, X[p], lane);
        gs_d4 R = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const double av = Dv[(i * 16 + fr) * GS_DV_STR + 4 * s4 + fq];                 // A operand: D_i
            R = __builtin_amdgcn_mfma_f64_16x16x4f64(av, T[s4], R, 0, 0, 0);
        }
        X[i] = R;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int x = 0; x < 4; ++x)
            Linv[(16 * i + fq + 4 * x) * 128 + 16 * J + fr] = (i < J) ? 0.0 : X[i][x];
}

// the whole 128 x 128 inverse: wave w builds block columns w and 7 - w
__device__ __forceinline__ void gs_trtri_block(const double* Ls, const double* Dv, double* Linv, int w, int lane) {
    if (w == 0) {
        gs_trtri_col2<0>(Ls, Dv, Linv, lane);
        gs_trtri_col2<7>(Ls, Dv, Linv, lane);
    } else if (w == 1) {
        gs_trtri_col2<1>(Ls, Dv, Linv, lane);
        gs_trtri_col2<6>(Ls, Dv, Linv, lane);
    } else if (w == 2) {
        gs_trtri_col2<2>(Ls, Dv, Linv, lane);
        gs_trtri_col2<5>(Ls, Dv, Linv, lane);
    } else {
        gs_trtri_col2<3>(Ls, Dv, Linv, lane);
        gs_trtri_col2<4>(Ls, Dv, Linv, lane);
    }
}

// The diagonal-block routine as a device function (k_potrf_diag, k_potrf_diag256, k_chain and the fused small / medium kernels
// share it).  A: the 128 x 128 block (leading dimension ld), factored in place (lower part).  diag0: the block's 128 original
// diagonal entries (pivot guard).  Returns 0 or the 1-based local column of the first bad pivot (uniform over the workgroup);
// *logdet_out (thread 0) = sum_j log L_jj.  wsp: the caller's LDS workspace (16-B aligned); passing it in lets a fused kernel lend
// the same bytes to its other phases.  What it leaves behind:
//   - the substitution tables of the block stay in the caller's LDS workspace, wsp[0 .. GS_LTAB): the 28 panel dumps
//     (-L_kj in A-operand layout) and the 8 micro-block inverses D_j.  gs_panel16 solves rows against them;
//   - Ltab != NULL: the same GS_LTAB doubles are copied to global memory for kernels that come later;
//   - Linv != NULL: the explicit 128 x 128 inverse is built too (phase 2; 15 k cycles that nothing on the
//     factorisation's own path needs any more).
#define GS_LTAB (GS_D2_DV + 8 * 16 * GS_DV_STR)          // 9344 doubles = 73 KB
// LDS layout of the !FULL mode: 8 panel-column slots | the 8 micro-block inverses | 128 thresholds   (4352 doubles = 34 KB)
#define GS_D2C_DV (8 * 256)
#define GS_D2C_THR (GS_D2C_DV + 8 * 16 * GS_DV_STR)
#define GS_D2C_WS (GS_D2C_THR + 128)
template <bool FULL = true, bool PARTIAL = false>
__device__ __forceinline__ int gs_diag_block(double* A, int64_t ld, double* Linv, double* Ltab, double* logdet_out,
                                             const double* diag0, unsigned long long* stamps, double* wsp, int nbv = 8) {
    __shared__ double dbuf[128];
    __shared__ int fail_sh;
    double* Ls = wsp + GS_D2_LS;
    double* Dv = wsp + (FULL ? GS_D2_DV : GS_D2C_DV);
    double* thr = wsp + (FULL ? GS_D2_THR : GS_D2C_THR);
    double* scr = nullptr;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    unsigned long long st0 = 0, sr0 = 0, st2 = 0;
    if (stamps) {
        st0 = __builtin_amdgcn_s_memtime();
        sr0 = __builtin_amdgcn_s_memrealtime();
        if (t == 0) stamps[7] = st0;
    }
    if (t == 0) fail_sh = -1;                           // first read behind the first barrier of step 0
    double d0 = 0.0;
    if (t < 128) d0 = diag0[t];                         // the thresholds' load goes out ahead of the block's (see gs_d2_wave)
    __builtin_amdgcn_sched_barrier(0);
    bool ok;
    if (w == 0) ok = gs_d2_wave<0, FULL, PARTIAL>(A, ld, Dv, scr, Ls, Ltab, thr, d0, dbuf, &fail_sh, lane, stamps, nbv);
    else if (w == 1) ok = gs_d2_wave<1, FULL, PARTIAL>(A, ld, Dv, scr, Ls, Ltab, thr, d0, dbuf, &fail_sh, lane, stamps, nbv);
    else if (w == 2) ok = gs_d2_wave<2, FULL, PARTIAL>(A, ld, Dv, scr, Ls, Ltab, thr, d0, dbuf, &fail_sh, lane, stamps, nbv);
    else ok = gs_d2_wave<3, FULL, PARTIAL>(A, ld, Dv, scr, Ls, Ltab, thr, d0, dbuf, &fail_sh, lane, stamps, nbv);
    if (!ok) return fail_sh + 1;                       // uniform: every wave read the flag behind the same barrier
    if (stamps) st2 = __builtin_amdgcn_s_memtime();
    if (t < 128) dbuf[t] = log(dbuf[t]);
    if constexpr (FULL) {
        if (Ltab) {
            const gs_d2* src = reinterpret_cast<const gs_d2*>(wsp);
            gs_d2* dst = reinterpret_cast<gs_d2*>(Ltab);
            for (int i = t; i < GS_LTAB / 2; i += 256) dst[i] = src[i];
        }
        if (Linv) gs_trtri_block(Ls, Dv, Linv, w, lane);
    } else {
        // the panel dumps went to Ltab as they were made; only the micro-block inverses are left to export
        const gs_d2* src = reinterpret_cast<const gs_d2*>(Dv);
        gs_d2* dst = reinterpret_cast<gs_d2*>(Ltab + GS_D2_DV);
        for (int i = t; i < 8 * 16 * GS_DV_STR / 2; i += 256) dst[i] = src[i];
    }
    __threadfence_block();
    __syncthreads();
    if (w == 0) {
        // sum of the 128 logs by one wave: two per lane, then a fixed xor tree (deterministic)
        double sl = dbuf[lane] + dbuf[lane + 64];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sl += __shfl_xor(sl, off, 64);
        if (lane == 0) *logdet_out = sl;
    }
    if (t == 0) {
        if (stamps) {
            const unsigned long long st3 = __builtin_amdgcn_s_memtime(), sr3 = __builtin_amdgcn_s_memrealtime();
            stamps[0] = 0;              // (the loads are part of phase 1 in this version)
            stamps[1] = st2 - st0;      // phase 1 (micro-block factorisation)
            stamps[2] = st3 - st2;      // table export (+ block inverse when asked for)
            stamps[3] = st3 - st0;      // total shader cycles
            stamps[4] = sr3 - sr0;      // total 100 MHz ticks
        }
    }
    return 0;
}

