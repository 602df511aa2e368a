// the bulk trailing-update tile (k_gemm_ld3: 128 x 64, LDS-direct staging), its grouped launches, the MFMA-rate probe kernel
// (part of gsum_kernels.hip.h: included from there, in order; gfx950 only)
#pragma once
template <int NACC>
__global__ __launch_bounds__(512) void k_mfma_peak(double* out, int iters) {
    gs_d4 acc[NACC];
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = gs_d4{0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (sum == 1.2345e301) out[blockIdx.x * blockDim.x + threadIdx.x] = sum;          // keeps the chain alive
}

// (body of k_gemm_ld3 / k_gemm_ld3g: `bid_in` is the tile's index within ITS product -- the workgroup id of a plain launch, the
// offset into its entry's tile range for a grouped one)
// Lower-triangle launches of the 128 x 64 tile without counted tiles (the batch's far updates, plain SYRKs): number of tiles of an M x M triangle.
// M a multiple of 128: row bm holds column tiles 0 .. 2 bm + 1.  Otherwise (round 5) the PARTIAL row tile comes FIRST: the trailing matrices of
// the bordered factorisation have 128 k + 16 rows (the right-hand-side rows), and with the partial tile row last, 2 tm + 2 tiles -- the longest
// row of the triangle: 114 of 3306 at M = 7184 -- multiply for 16 valid rows each.  Rows [0, off) (off = M mod 128) form tile row 0 (c0 column
// tiles), row bm >= 1 covers [off + 128 (bm - 1), off + 128 bm) and holds 2 bm + c0 column tiles: 3249 tiles at M = 7184.
__host__ __device__ inline bool gs_tri_shifted(int M) { return (M & 127) != 0 && M > 128; }
__host__ __device__ inline int64_t gs_tri_tiles64(int64_t M) {
    const int64_t tm = (M + 127) / 128;
    if (!gs_tri_shifted((int)M)) return tm * (tm + 1);
    const int64_t off = M & 127, c0 = (off - 1) / 64 + 1, R = 1 + M / 128;
    return c0 + (R - 1) * (R + c0);
}

// ... with COUNTED tiles (the single-factorisation schedules: the tiles of the first 256 columns first, counted for the chain kernel; then,
// in the deep schedule, column tiles 4 .. c2 - 1): how many tiles each group holds, for the host's wait counts and the kernel's id -> tile map.
// Row bm holds nt(bm) column tiles: 2 bm + 2 unshifted; shifted c0 for row 0 and 2 bm + c0 for bm >= 1 (c0 = 1 or 2).
__host__ __device__ inline int gs_tri_row_tiles(int M, int bm) {
    if (!gs_tri_shifted(M)) return 2 * bm + 2;
    const int c0 = ((M & 127) - 1) / 64 + 1;
    return bm == 0 ? c0 : 2 * bm + c0;
}
__host__ __device__ inline int gs_tri_rows(int M) { return gs_tri_shifted(M) ? 1 + M / 128 : (M + 127) / 128; }
// largest q >= 0 with q (q + e - 1) <= f  (e = 1 or 2: tiles before row q of a run of rows holding e, e + 2, e + 4, ... tiles)
__device__ __forceinline__ int gs_tri_q(int f, int e) {
    int q = (int)((-(double)(e - 1) + sqrt((double)(e - 1) * (e - 1) + 4.0 * (double)f)) * 0.5);
    if (q < 0) q = 0;
    while ((int64_t)q * (q + e - 1) > f) --q;
    while ((int64_t)(q + 1) * (q + e) <= f) ++q;
    return q;
}
__host__ __device__ inline unsigned gs_tri_first4(int64_t M) {               // tiles with column tile < 4
    const int R = gs_tri_rows((int)M);
    unsigned cnt = 0;
    for (int bm = 0; bm < R && bm < 2; ++bm) cnt += (unsigned)(gs_tri_row_tiles((int)M, bm) < 4 ? gs_tri_row_tiles((int)M, bm) : 4);
    return cnt + (R > 2 ? 4u * (unsigned)(R - 2) : 0u);
}
__host__ __device__ inline unsigned gs_tri_second(int64_t M, int c2) {       // tiles with 4 <= column tile < c2
    const int R = gs_tri_rows((int)M);
    unsigned cnt = 0;
    for (int bm = 2; bm < R; ++bm) {
        const int nt = gs_tri_row_tiles((int)M, bm), hi = nt < c2 ? nt : c2;
        if (hi > 4) cnt += (unsigned)(hi - 4);
        if (nt >= c2) { cnt += (unsigned)(R - 1 - bm) * (unsigned)(c2 - 4); break; }      // every later row holds all c2 - 4
    }
    return cnt;
}

// BNT = 1: the 128 x 64 workgroup tile (8 waves of 32 x 32, 63 registers, three workgroups per CU) -- every launch of rounds 2-4.
// BNT = 2 (round 5): 128 x 128 (8 waves of 64 x 32: WM = 4, 2 waves down x 4 across; ~100 registers, 66 KB of LDS, two workgroups per CU): 0.75
// LDS reads and 0.125 LDS-direct loads per MFMA instead of 1 and 0.1875, a third fewer operand bytes per flop.  Same k order per accumulator:
// bit-identical.  Chosen per launch for large trailing matrices (gs_gemm / gs_lml_wave); the counted-tile orders exist for BNT = 1 only.
template <int NST, int BNT = 1>
__device__ __forceinline__ void gs_gemm_ld3_body(double* C, int64_t ldc, const double* A, int64_t lda, const double* B,
                                                 int64_t ldb, int M, int N, int K, int tri, int beta, double sign,
                                                 unsigned long long* kst, int nfirst, unsigned* first_done, const int bid_in,
                                                 const int c2 = 0, unsigned* second_done = nullptr) {
    constexpr int WM = BNT == 2 ? 4 : 2, WN = 2, WAVES_M = BNT == 2 ? 2 : 4, BM = 128, BN = 64 * BNT;
    static_assert(BNT == 1 || (BNT == 2 && NST == 2), "the 128 x 128 tile exists with two LDS stages only");
    constexpr int OPA = BM * GS_KC + 2, OPB = BN * GS_KC + 2, STAGE = OPA + OPB;
    constexpr int HALFA = BM / 2 * GS_KC + 1, HALFB = BN / 2 * GS_KC + 1;
    extern __shared__ double lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    if (kst && t == 0) atomicMin(kst, __builtin_amdgcn_s_memrealtime());             // diagnostics: first start / last end of the launch
    // (Tried in round 3, measured, not kept: de-phasing naps for the second and third workgroup of a CU -- no gain, the
    // co-resident workgroups are not in lock step; three LDS stages with two workgroups per CU -- 3 % slower alone, 7 % slower
    // pipelined.  A K = 256 launch at M = 7936 spends ~30-45 us on its C reads and ~25 us on its C stores of ~300; the K loop alone
    // runs at 66 TF/s, the clock-limited rate under this kernel: profiles/r03_bulk_cphase.log.)
    const int wm = w % WAVES_M, wn = w / WAVES_M;
    int bm, bn;
    bool first_cols = false, second_cols = false, shifted = false;
    if (BNT == 2 && tri) {
        // square tiles: row bm of the lower triangle holds column tiles 0 .. bm
        const int bid = bid_in;
        bm = (int)((sqrt(8.0 * (double)bid + 1.0) - 1.0) * 0.5);
        while ((int64_t)(bm + 1) * (bm + 2) / 2 <= bid) ++bm;
        while ((int64_t)bm * (bm + 1) / 2 > bid) --bm;
        bn = bid - (int)((int64_t)bm * (bm + 1) / 2);
    } else if (tri == 1 && nfirst > 0 && BNT == 1 && gs_tri_shifted(M)) {
        // counted tiles on the shifted grid (partial row tile first; row 0 holds c0 column tiles, row bm >= 1 holds 2 bm + c0, c0 = 1 or 2):
        // group 1 = column tiles < 4 of every row, group 2 (c2 > 4) = column tiles 4 .. c2 - 1, then the rest.  Closed forms: a run of rows
        // whose counts grow by two per row, e + 2 q in row q of the run, has q (q + e - 1) tiles before row q (gs_tri_q inverts that);
        // the host counts the groups with gs_tri_first4 / gs_tri_second, which walk the same per-row counts.
        const int R = gs_tri_rows(M), c0 = ((M & 127) - 1) / 64 + 1;
        int f = bid_in;
        shifted = true;
        if (f < nfirst) {
            const int n0r = c0 < 4 ? c0 : 4, n1r = 2 + c0 < 4 ? 2 + c0 : 4;          // rows 0 and 1 hold fewer than four
            if (f < n0r) { bm = 0; bn = f; }
            else if (f < n0r + n1r) { bm = 1; bn = f - n0r; }
            else { bm = 2 + (f - n0r - n1r) / 4; bn = (f - n0r - n1r) % 4; }
            first_cols = true;
        } else {
            f -= nfirst;
            const int lo = c2 > 4 ? c2 : 4;
            const int nsec = c2 > 4 ? (int)gs_tri_second(M, c2) : 0;
            if (f < nsec) {
                // rows 2 .. bmF - 1 hold c0 + 2 (bm - 2) tiles of the group, rows from bmF = ceil((c2 - c0) / 2) on all c2 - 4
                const int bmF = (c2 - c0 + 1) / 2, qF = bmF - 2, cumF = qF > 0 ? qF * (qF + c0 - 1) : 0;
                if (f < cumF) {
                    const int q = gs_tri_q(f, c0);
                    bm = 2 + q;
                    bn = 4 + f - q * (q + c0 - 1);
                } else {
                    bm = bmF + (f - cumF) / (c2 - 4);
                    bn = 4 + (f - cumF) % (c2 - 4);
                }
                second_cols = true;
            } else {
                f -= nsec;
                // the rest: rows from bmS = floor((lo - c0) / 2) + 1 on hold 2 bm + c0 - lo = e + 2 (bm - bmS) tiles from column tile lo on
                const int bmS = (lo - c0) / 2 + 1, e = 2 * bmS + c0 - lo;
                const int q = gs_tri_q(f, e);
                bm = bmS + q;
                bn = lo + f - q * (q + e - 1);
            }
        }
        if (bm >= R) return;                     // (cannot happen: the grid is gs_tri_tiles64(M))
    } else if (tri && nfirst > 0) {
        // row bm of the lower triangle holds column tiles 0 .. 2 bm + 1 (64 wide); the first four of every row come first
        // (row 0 has two), then rows 2.. with their tiles 4 .. 2 bm + 1
        const int bid = bid_in;
        if (bid < nfirst) {
            if (bid < 2) { bm = 0; bn = bid; }
            else { bm = 1 + (bid - 2) / 4; bn = (bid - 2) % 4; }
            first_cols = true;
        } else if (c2 > 4) {
            // A SECOND counted group (the deep single-factorisation schedule, gs_potrf_chain): column tiles 4 .. c2 - 1 of every row that
            // has them -- the columns of the next macro-step's panels -- come right behind the first four and are counted in
            // *second_done; then the rest.  Row bm holds column tiles 0 .. 2 bm + 1: rows 2 .. c2 / 2 - 2 hold 2 (bm - 1) tiles of the
            // group ((bm - 1)(bm - 2) before row bm), rows from bm0 = c2 / 2 - 1 on all c2 - 4.
            int f = bid - nfirst;
            const int tmr = (M + BM - 1) / BM, bm0 = c2 / 2 - 1, cum0 = (bm0 - 1) * (bm0 - 2);
            const int nsecond = tmr <= bm0 ? (tmr >= 2 ? (tmr - 1) * (tmr - 2) : 0) : cum0 + (tmr - bm0) * (c2 - 4);
            if (f < nsecond) {
                if (f < cum0) {
                    bm = (int)((3.0 + sqrt(1.0 + 4.0 * (double)f)) * 0.5);
                    while ((int64_t)(bm - 1) * (bm - 2) > f) --bm;
                    while ((int64_t)bm * (bm - 1) <= f) ++bm;
                    bn = 4 + f - (bm - 1) * (bm - 2);
                } else {
                    bm = bm0 + (f - cum0) / (c2 - 4);
                    bn = 4 + (f - cum0) % (c2 - 4);
                }
                second_cols = true;
            } else {
                f -= nsecond;                                   // rows bm = c2 / 2 + q hold 2 (q + 1) tiles from c2 on: q (q + 1) before them
                int q = (int)((sqrt(1.0 + 4.0 * (double)f) - 1.0) * 0.5);
                while ((int64_t)q * (q + 1) > f) --q;
                while ((int64_t)(q + 1) * (q + 2) <= f) ++q;
                bm = c2 / 2 + q;
                bn = c2 + f - q * (q + 1);
            }
        } else {
            const int f = bid - nfirst;
            bm = (int)((3.0 + sqrt(1.0 + 4.0 * (double)f)) * 0.5);
            while ((int64_t)(bm - 1) * (bm - 2) > f) --bm;
            while ((int64_t)bm * (bm - 1) <= f) ++bm;
            bn = 4 + f - (bm - 1) * (bm - 2);
        }
    } else if (tri == 1 && BNT == 1 && gs_tri_shifted(M)) {
        // the partial row tile first (gs_tri_tiles64): tile row 0 = rows [0, off), c0 column tiles; row bm >= 1 holds 2 bm + c0
        const int off = M & 127, c0 = (off - 1) / 64 + 1;
        const int bid = bid_in;
        if (bid < c0) {
            bm = 0;
            bn = bid;
        } else {
            const int g = bid - c0;
            bm = (int)((-(double)(c0 - 1) + sqrt((double)(c0 - 1) * (c0 - 1) + 4.0 * (double)(c0 + g))) * 0.5);
            if (bm < 1) bm = 1;
            while ((int64_t)(bm - 1) * (bm + c0) > g) --bm;
            while ((int64_t)bm * (bm + 1 + c0) <= g) ++bm;
            bn = g - (bm - 1) * (bm + c0);
        }
        shifted = true;
    } else if (tri) {
        // lower tiles of a square C with 128 x 64 tiles: row bm holds column tiles 0 .. 2 bm + 1.
        // (An XCD-aware order -- rows padded to multiples of 8 slots so that workgroup id and column tile agree modulo 8
        // and each XCD's L2 keeps one eighth of the B-side panel -- was measured: rocprofv3 FETCH_SIZE of the exclusive
        // M = 8192 launch 813 -> 610 MB, its rate unchanged (55.0 vs 55.6 TF/s), the 16-in-flight pipeline 3 % SLOWER
        // (266.6 vs 274 evals/s: with sixteen queues dispatching at once workgroup ids no longer map to XCDs round-robin,
        // and the padding slots cost launches).  The kernel is not fetch-bound; the plain order stays.)
        const int bid = bid_in;
        bm = (int)((sqrt(4.0 * (double)bid + 1.0) - 1.0) * 0.5);
        while ((int64_t)(bm + 1) * (bm + 2) <= bid) ++bm;
        while ((int64_t)bm * (bm + 1) > bid) --bm;
        bn = bid - (int)((int64_t)bm * (bm + 1));
    } else {
        const int tm = (M + BM - 1) / BM;
        bm = bid_in % tm;
        bn = bid_in / tm;
        first_cols = bid_in < nfirst;        // column-major tile order: the first 4 tm ids are the first 256 columns
    }
    // (shifted grid: tile row 0 ends at row off -- the rows behind it belong to tile row 1 -- and the full tile rows start there)
    const int m0 = shifted ? (bm == 0 ? 0 : (M & 127) + (bm - 1) * BM) : bm * BM, n0 = bn * BN;
    const int Mr = (shifted && bm == 0) ? (M & 127) : M;          // row bound of THIS tile
    if (n0 >= N) {                            // tri: the last row of a ragged matrix may have one column tile too many
        if (first_cols && t == 0) gs_flag_add(first_done);
        if (second_cols && t == 0) gs_flag_add(second_done);
        return;
    }
    if (tri == 2) {
        A += m0;
        B += m0;
        K -= m0;
    }
    const int fr = lane & 15, fq = lane >> 4;
    const bool neg = sign < 0.0;
    gs_d4 acc[WM][WN];
    // Lower-triangle launches: a wave whose 32 x 32 block lies strictly ABOVE the diagonal (5 of the 8 waves of a diagonal block's
    // right-hand tile, 1 of 8 in its left-hand tile) neither loads, multiplies nor stores it -- nothing reads the strict upper triangle
    // of a workspace matrix -- and leaves the matrix pipes to the CU's other workgroups: 0.75 / (tm + 1) of a launch's MFMAs (1.3 % at
    // tm = 57 row tiles, 4.4 % at 16).  It still stages its share of the operands and keeps the barriers.
    const bool idle = tri == 1 && __builtin_amdgcn_readfirstlane(m0 + wm * WM * 16 + WM * 16 - 1 < n0 + wn * WN * 16);
    // interior tiles (all but the last row / column of a ragged matrix): the 16 C loads -- and the 16 stores at the end -- go out back to back, without a
    // compare and a branch each (same-process A/B, profiles/r03_bulk_interior_tiles_ab.log: batch +0.8 %, K = 256 / 512 steady state at M = 7936 +2 / +1.5 %,
    // M = 4096 -1.1 %, one factorisation unchanged; bit-identical)
    const bool full = m0 + BM <= Mr && n0 + BN <= N;
    if (idle) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j] = gs_d4{0.0, 0.0, 0.0, 0.0};
    } else if (full && beta) {
        const double* c0 = C + (int64_t)(m0 + wm * WM * 16 + fq) * ldc + n0 + wn * WN * 16 + fr;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int x = 0; x < 4; ++x) acc[i][j][x] = c0[(int64_t)(16 * i + 4 * x) * ldc + 16 * j];
    } else {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = n0 + (wn * WN + j) * 16 + fr;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int row = m0 + (wm * WM + i) * 16 + fq + 4 * x;
                acc[i][j][x] = (beta && row < Mr && col < N) ? C[(int64_t)row * ldc + col] : 0.0;      // sign applied below, behind the wait
            }
        }
    }
    // staging: A has 16 eight-row slices (2 per wave: rows [16 w, 16 w + 16) by parity), B has 8 (1 per wave: wave w
    // takes parity w & 1 of rows [16 (w >> 1), 16 (w >> 1) + 16))
    const int lrow = lane >> 3, lg = lane & 7;
    const double* srcA[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = 16 * w + 2 * lrow + h;
        const int kp = lg ^ ((r >> 1) & 7);
        int ra = m0 + r;
        ra = ra < Mr ? ra : Mr - 1;
        srcA[h] = A + (int64_t)ra * lda + 2 * kp;
    }
    const int hb = w & 1, gb = w >> 1;
    const double* srcB;
    const double* srcB2[2];                 // (BNT == 2: B has 16 eight-row slices too, two per wave, staged exactly like A)
    if (BNT == 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = 16 * w + 2 * lrow + h;
            const int kp = lg ^ ((r >> 1) & 7);
            int rb = n0 + r;
            rb = rb < N ? rb : N - 1;
            srcB2[h] = B + (int64_t)rb * ldb + 2 * kp;
        }
        srcB = srcB2[0];
    } else {
        const int r = 16 * gb + 2 * lrow + hb;
        const int kp = lg ^ ((r >> 1) & 7);
        int rb = n0 + r;
        rb = rb < N ? rb : N - 1;
        srcB = B + (int64_t)rb * ldb + 2 * kp;
        srcB2[0] = srcB2[1] = srcB;
    }
    auto stage_load = [&](int kc, int stage) {
        double* base = lds + stage * STAGE;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            __builtin_amdgcn_global_load_lds(srcA[h] + kc * GS_KC, base + h * HALFA + 8 * w * GS_KC, 16, 0, 0);
        if (BNT == 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                __builtin_amdgcn_global_load_lds(srcB2[h] + kc * GS_KC, base + OPA + h * HALFB + 8 * w * GS_KC, 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds(srcB + kc * GS_KC, base + OPA + hb * HALFB + 8 * gb * GS_KC, 16, 0, 0);
        }
    };
    const int swz = (fr >> 1) & 7;
    const int rselA = (fr & 1) * HALFA + (fr >> 1) * GS_KC, rselB = (fr & 1) * HALFB + (fr >> 1) * GS_KC;
    int goff[GS_KC / 4];
#pragma unroll
    for (int ks = 0; ks < GS_KC / 4; ++ks) goff[ks] = (((2 * ks + (fq >> 1)) ^ swz) << 1) + (fq & 1);
    const int nk = K / GS_KC;
    // NST stages of LDS: chunk c + NST - 1 is requested while chunk c is multiplied.  Every wave issues exactly three
    // LDS-direct loads per stage, so "all but the newest NST - 2 stages have landed" is vmcnt(3 (NST - 2)).
    // (NST = 3: 74 KB per workgroup, two per CU; the operands of a K = 256 trailing update mostly MISS the L2 -- the panel is
    // 16 MB, FETCH_SIZE ~ the operand bytes -- and come from the Infinity Cache in 1-2 us, more than one chunk of a shared CU.)
    stage_load(0, 0);
    if (NST == 3 && nk > 1) stage_load(1, 1);
    // The C values were requested first and are used (negated) only from here on: with the negation next to the loads the compiler put its
    // wait for them in front of the third LDS-direct load of stage 0, which then paid a memory latency of its own in every tile's prologue.
    // (same-process A/B, profiles/r03_bulk_prologue_ab.log: +1 % on the kernel at K = 256, +0.2-0.3 % on the pipelined batch; bit-identical)
    __builtin_amdgcn_sched_barrier(0);
    if (NST == 3 && nk > 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (neg) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int x = 0; x < 4; ++x) acc[i][j][x] = -acc[i][j][x];
    }
    __syncthreads();
    if constexpr (NST == 2) {
        // Fragments one k-step ahead, in two register sets: the LDS reads of step ks + 1 are issued BEFORE the four MFMAs of step ks, and the last
        // step of a chunk is multiplied behind the barrier, after the next chunk's loads and first reads have gone out -- a wave never sits between "MFMAs issued" and
        // "next fragments arrived" with nothing to issue (the compiler's own schedule reused one register set: read, wait, multiply, four times per
        // chunk).  Same products, same order per accumulator: bit-identical.  +7 registers (63 of the 72 this kernel may use).  Same-process A/B
        // (profiles/r03_bulk_kloop_ab.log): pipelined batch +1.25 % (294.4-295.0 against 290.8-291.5 evals/s), K = 512 steady state +1.6 %, M = 4096 +1.2 %,
        // one factorisation -0.9 % time; K = 256 at M = 7936 unchanged (that launch is held by its C phases).
        static_assert(GS_KC == 16, "the pipelined K loop is written for four k-steps per chunk");
        double afA[WM], bfA[WN], afB[WM], bfB[WN];
        auto ldf = [&](double (&af)[WM], double (&bf)[WN], const double* sA, const double* sB, int ks) {
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = sA[i * 8 * GS_KC + goff[ks]];
#pragma unroll
            for (int j = 0; j < WN; ++j) bf[j] = sB[j * 8 * GS_KC + goff[ks]];
        };
        auto mm = [&](const double (&af)[WM], const double (&bf)[WN]) {
            if (idle) return;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        };
        const double* sA = lds + wm * WM * 8 * GS_KC + rselA;
        const double* sB = lds + OPA + wn * WN * 8 * GS_KC + rselB;
        if (nk > 1) stage_load(1, 1);
        if (nk > 0) ldf(afA, bfA, sA, sB, 0);
        // `arrived` is an empty statement that reads a fragment set: the compiler's wait for that set lands THERE, i.e. before the next set's reads are
        // issued -- placed in front of the MFMAs (its own choice) the wait came out as lgkmcnt(0) and covered the reads just issued as well.
        auto arrived = [&](const double (&af)[WM], const double (&bf)[WN]) {
            static_assert((WM == 2 || WM == 4) && WN == 2, "two or four row and two column fragments per wave");
            if constexpr (WM == 4) asm volatile("" ::"v"(af[0]), "v"(af[1]), "v"(af[2]), "v"(af[3]), "v"(bf[0]), "v"(bf[1]));
            else asm volatile("" ::"v"(af[0]), "v"(af[1]), "v"(bf[0]), "v"(bf[1]));
        };
        for (int c = 0; c < nk; ++c) {
            __builtin_amdgcn_sched_barrier(0);
            arrived(afA, bfA);
            ldf(afB, bfB, sA, sB, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(afA, bfA);
            __builtin_amdgcn_sched_barrier(0);
            arrived(afB, bfB);
            ldf(afA, bfA, sA, sB, 2);
            __builtin_amdgcn_sched_barrier(0);
            mm(afB, bfB);
            __builtin_amdgcn_sched_barrier(0);
            arrived(afA, bfA);
            ldf(afB, bfB, sA, sB, 3);
            __builtin_amdgcn_sched_barrier(0);
            mm(afA, bfA);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                // (waits for the LDS reads above too: this stage may be overwritten from here on)
            if (c + 1 < nk) {                               // the next chunk's first fragments are requested BEFORE the last four MFMAs of this one ...
                sA = lds + ((c + 1) & 1) * STAGE + wm * WM * 8 * GS_KC + rselA;
                sB = lds + ((c + 1) & 1) * STAGE + OPA + wn * WN * 8 * GS_KC + rselB;
                ldf(afA, bfA, sA, sB, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mm(afB, bfB);                                   // k-step 3 of chunk c
            // ... and the LDS-direct loads of chunk c + 2 behind them: their address arithmetic and M0 writes no longer stand between the barrier and
            // the MFMAs (same-process A/B, profiles/r03_bulk_dma_order_ab.log: batch +0.7 %, M = 4096 +2.2 %, K = 256 / 512 steady state +1.5 / +0.7 %;
            // one MFMA group later still is no better and costs M = 4096 1.7 %).
            __builtin_amdgcn_sched_barrier(0);
            if (c + 2 < nk) stage_load(c + 2, c & 1);
        }
    } else {
    for (int c = 0; c < nk; ++c) {
        if (c + NST - 1 < nk) stage_load(c + NST - 1, (c + NST - 1) % NST);
        const double* sA = lds + (c % NST) * STAGE + wm * WM * 8 * GS_KC + rselA;
        const double* sB = lds + (c % NST) * STAGE + OPA + wn * WN * 8 * GS_KC + rselB;
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
        if (NST == 3 && c + 2 < nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");       // chunk c + 1 has landed; c + 2 may be in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    }
    // The store addresses are re-derived from the thread index HERE, behind an opaque copy of it: computed once in the prologue they
    // stay live across the K loop, and this kernel must fit 72 VGPRs -- six bulk waves then leave a SIMD exactly the room in which one
    // chain / panel wave (224) fits as soon as ONE bulk workgroup retires (tests/test_host_logic.py::test_kernel_register_budgets).
    int t2 = threadIdx.x;
    asm volatile("" : "+v"(t2));
    const int lane2 = t2 & 63, w2 = t2 >> 6;
    const int fr2 = lane2 & 15, fq2 = lane2 >> 4, wm2 = w2 % WAVES_M, wn2 = w2 / WAVES_M;
    if (idle) {
        // (nothing to store)
    } else if (!(first_cols || second_cols) && m0 + BM <= Mr && n0 + BN <= N) {
        double* c0 = C + (int64_t)(m0 + wm2 * WM * 16 + fq2) * ldc + n0 + wn2 * WN * 16 + fr2;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int x = 0; x < 4; ++x) c0[(int64_t)(16 * i + 4 * x) * ldc + 16 * j] = neg ? -acc[i][j][x] : acc[i][j][x];
    } else if (!(first_cols || second_cols)) {      // (counted tiles: write-through stores below -- their readers do not wait for this launch to end)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int col = n0 + (wn2 * WN + j) * 16 + fr2;
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int row = m0 + (wm2 * WM + i) * 16 + fq2 + 4 * x;
                    if (row < Mr && col < N) C[(int64_t)row * ldc + col] = neg ? -acc[i][j][x] : acc[i][j][x];
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int col = n0 + (wn2 * WN + j) * 16 + fr2;
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int row = m0 + (wm2 * WM + i) * 16 + fq2 + 4 * x;
                    if (row < Mr && col < N) gs_st_wt(C + (int64_t)row * ldc + col, neg ? -acc[i][j][x] : acc[i][j][x]);
                }
            }
    }
    if (first_cols || second_cols) {          // published to the chain kernel / the near stream: every wave drains, then one lane counts the tile
        gs_drain();
        __syncthreads();
        if (t == 0) gs_flag_add(first_cols ? first_done : second_done);
    }
    if (kst && t == 0) atomicMax(kst + 1, __builtin_amdgcn_s_memrealtime());
}


template <int NST>
__global__ __launch_bounds__(512, NST == 2 ? 7 : 4) void k_gemm_ld3(double* C, int64_t ldc, const double* A, int64_t lda, const double* B,
                                                      int64_t ldb, int M, int N, int K, int tri, int beta, double sign,
                                                      unsigned long long* kst, int nfirst, unsigned* first_done, int c2, unsigned* second_done) {
    gs_gemm_ld3_body<NST>(C, ldc, A, lda, B, ldb, M, N, K, tri, beta, sign, kst, nfirst, first_done, (int)blockIdx.x, c2, second_done);
}

// the 128 x 128 tile (BNT = 2) as a plain launch: lower triangles / rectangles without counted tiles
__global__ __launch_bounds__(512, 4) void k_gemm_ld3b(double* C, int64_t ldc, const double* A, int64_t lda, const double* B, int64_t ldb, int M, int N,
                                                      int K, int tri, int beta, double sign, unsigned long long* kst) {
    gs_gemm_ld3_body<2, 2>(C, ldc, A, lda, B, ldb, M, N, K, tri, beta, sign, kst, 0, (unsigned*)nullptr, (int)blockIdx.x);
}

// ---- grouped launches: the same outer step of SEVERAL evaluations in one launch -------------------------------------------------
// A batch of evaluations (a likelihood grid) used to run as up to 20 independent HIP streams, one evaluation each, and counted on
// the runtime giving every stream a hardware queue of its own (GPU_MAX_HW_QUEUES) -- 24 live streams collapsed it, a process with an
// RCCL communicator had to run a different policy, and no per-launch profile described the step.  Round 4: the evaluations of a
// group advance in lock step and ONE launch carries the tiles of all of them.  An entry names its product by offsets from its
// evaluation's workspace (all workspaces of a group are `strideA` doubles apart, same order, same leading dimension); the entry of a
// workgroup is found from the running tile counts in the kernel arguments (scalar loads: the workgroup id is uniform).  The tile
// arithmetic is gs_gemm_ld3_body's: results are bit-identical to the one-evaluation launches.
#define GS_WV_MAX 24
struct gs_wv_gemm_entry {
    int64_t offC, offA, offB;      // doubles from the evaluation's workspace base
    int M, N, K, tri;
    int q, pad;                    // workspace index within the group
};
struct gs_wv_gemm_args {
    double* base; int64_t strideA, ld;
    int n, pad;
    int end[GS_WV_MAX];            // running tile counts: entry e owns block ids [end[e - 1], end[e])
    gs_wv_gemm_entry e[GS_WV_MAX];
};
__device__ __forceinline__ void gs_gemm_ld3g_body(const gs_wv_gemm_args& a) {
    const int bid = (int)blockIdx.x;
    int e = 0;
    while (e + 1 < a.n && bid >= a.end[e]) ++e;
    const int first = e ? a.end[e - 1] : 0;
    const gs_wv_gemm_entry& en = a.e[e];
    double* W = a.base + (int64_t)en.q * a.strideA;
    gs_gemm_ld3_body<2>(W + en.offC, a.ld, W + en.offA, a.ld, W + en.offB, a.ld, en.M, en.N, en.K, en.tri, 1, -1.0,
                        (unsigned long long*)nullptr, 0, (unsigned*)nullptr, bid - first);
}
// k_gemm_ld3g: the big ("far") trailing updates, one after the other on the batch schedule's bulk stream.  k_gemm_ld3n: the same
// code under a name of its own for the small "near" updates that run on the groups' chain streams BESIDE them -- so that a
// per-kernel profile (rocprofv3 --stats) keeps the two roles apart and the far updates' launch times add up to the step time.
template <int BNT>
__device__ __forceinline__ void gs_gemm_ld3g_body_t(const gs_wv_gemm_args& a) {
    const int bid = (int)blockIdx.x;
    int e = 0;
    while (e + 1 < a.n && bid >= a.end[e]) ++e;
    const int first = e ? a.end[e - 1] : 0;
    const gs_wv_gemm_entry& en = a.e[e];
    double* W = a.base + (int64_t)en.q * a.strideA;
    gs_gemm_ld3_body<2, BNT>(W + en.offC, a.ld, W + en.offA, a.ld, W + en.offB, a.ld, en.M, en.N, en.K, en.tri, 1, -1.0,
                             (unsigned long long*)nullptr, 0, (unsigned*)nullptr, bid - first);
}
// the far updates of large trailing matrices on the 128 x 128 tile (option wave_tile128_rows)
__global__ __launch_bounds__(512, 4) void k_gemm_ld3g2(const gs_wv_gemm_args a) { gs_gemm_ld3g_body_t<2>(a); }
__global__ __launch_bounds__(512, 7) void k_gemm_ld3g(const gs_wv_gemm_args a) { gs_gemm_ld3g_body(a); }
__global__ __launch_bounds__(512, 7) void k_gemm_ld3n(const gs_wv_gemm_args a) { gs_gemm_ld3g_body(a); }

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

// the same for the finished evaluations of a group (one workgroup per entry)
__global__ __launch_bounds__(256) void k_finalize_g(const gs_wv_chain_args a) {
    const int64_t q = a.q[blockIdx.x];
    const int t = threadIdx.x;
    const int r = t >> 4, c = t & 15;
    const double* A = a.p.A + q * a.p.strideA;
    double* res = a.p.res + q * 258;
    res[t] = -A[(int64_t)(a.p.np + r) * a.p.ld + a.p.np + c];
    if (t == 0) {
        double s = 0.0;
        for (int i = 0; i < a.p.T; ++i) s += a.p.logdet[q * a.p.T + i];
        res[256] = s;
        res[257] = (double)a.p.info[q];
    }
}

// out[r] = sum_j B[r][j]^2 over ncols; one wave per row, fixed summation order.
// Row sums of squares of B (nrows x ncols) AND, in the same pass over B, its product with the 16 rows of W (ldw apart):
//   ss[row] = sum_j B[row][j]^2,   vw[row * 16 + c] = sum_j B[row][j] W[c][j].
// predict reads V^T (m x n, 268 MB at m = 2048, n = 16384) for both: as k_rowsumsq + a 16-column GEMM on the 32 x 128 tile
// (64 workgroups looping over K = n: 0.79 ms, latency-bound) that was two passes and 0.9 ms; this is one streaming pass.
// One wave per GS_VW_ROWS rows (W -- 2 MB, L2-resident -- is re-read once per wave, not once per row).
#define GS_VW_ROWS 2
__global__ __launch_bounds__(256) void k_rowsumsq_vw(const double* B, int64_t ldb, int nrows, int ncols, const double* W, int64_t ldw,
                                                     double* ss, double* vw) {
    constexpr int R = GS_VW_ROWS;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= nrows) return;
    const double* p[R];
#pragma unroll
    for (int r = 0; r < R; ++r) p[r] = B + (int64_t)(row0 + r < nrows ? row0 + r : nrows - 1) * ldb;
    double s[R], acc[R][16];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s[r] = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[r][c] = 0.0;
    }
    for (int j = lane; j < ncols; j += 64) {
        double b[R], wv[16];
#pragma unroll
        for (int r = 0; r < R; ++r) b[r] = p[r][j];
#pragma unroll
        for (int c = 0; c < 16; ++c) wv[c] = W[(int64_t)c * ldw + j];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            s[r] += b[r] * b[r];
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[r][c] += b[r] * wv[c];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s[r] += __shfl_down(s[r], off, 64);
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[r][c] += __shfl_down(acc[r][c], off, 64);
        if (lane == 0 && row0 + r < nrows) {
            ss[row0 + r] = s[r];
#pragma unroll
            for (int c = 0; c < 16; ++c) vw[(int64_t)(row0 + r) * 16 + c] = acc[r][c];
        }
    }
}

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

