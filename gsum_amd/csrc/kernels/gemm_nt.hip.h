// K2b: C (+)= s A B^T with register staging (k_gemm_nt): chain GEMMs, border rows, sweeps
// (part of gsum_kernels.hip.h: included from there, in order; gfx950 only)
#pragma once
// ------------------------------------------------------------------------------------------------
// K2b: C (+)= sign * A * B^T  on fp64 MFMA.   A: M x K, B: N x K (both row-major, K contiguous — the
// shape every step of a row-major lower Cholesky produces), C: M x N.
//   - 4 waves per workgroup, wave tile (WM*16) x (WN*16) of v_mfma_f64_16x16x4_f64 accumulators;
//   - K is staged 16 doubles (one 128-B line per row) at a time: global -> registers -> LDS, two LDS
//     stages, one barrier per chunk; the next chunk's global loads are in flight during the MFMAs;
//   - fragment reads at row stride 17 doubles: conflict-free for the A/B lane map (lane l holds
//     [row l&15][k l>>4]) under ds_read2_b64's 32-bank mapping (stride 18 measured 40% conflict cycles);
//   - rows >= M / cols >= N are clamped on load and predicated on store, so the 16-row border tile
//     and the padded tail run through the same code;
//   - tri != 0: only tiles on or below the diagonal (SYRK of the trailing matrix);
//   - sign must be +1 or -1 (it multiplies the staged A operand exactly).
// In-place use (C == A, TRSM against an explicit inverse) is safe when one tile spans all N = K
// columns: every global load of the tile's rows is finished before the epilogue stores.
// ------------------------------------------------------------------------------------------------
// PF: operand chunks requested ahead of the one being multiplied.  1 = the next chunk only (one memory latency per 16
// columns of K: fine when several workgroups share a CU, 1.3-1.5 us per chunk for the lone 32 x 128 tiles of the
// factorisation's chain -- sibling update 12.6 us, look-ahead update 21 us for 0.4 / 0.9 us of MFMA work per tile).
// 4 = a ring of four register sets (K a multiple of 64): the same products in the same order, ~3x sooner.
template <int WM, int WN, int WAVES_M, int WAVES_N, bool STAMP = false, int PF = 1>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N == 8) ? 4 : 2) void k_gemm_nt(double* C, int64_t ldc, const double* A, int64_t lda,
                                                     const double* B, int64_t ldb, int M, int N, int K,
                                                     int tri, int beta, double sign,
                                                     unsigned long long* stamps = nullptr, int stagger = 0) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;          // 256 threads
    constexpr int BM = WM * 16 * WAVES_M, BN = WN * 16 * WAVES_N;
    constexpr int A_VECS = BM * (GS_KC / 2), B_VECS = BN * (GS_KC / 2);
    constexpr int A_IT = (A_VECS + NT - 1) / NT, B_IT = (B_VECS + NT - 1) / NT;
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
    if (tri == 2) {
        // C = U U^T for an upper-triangular U (row i is zero left of column i): a tile whose rows start at m0
        // only needs k >= m0 (m0 <= the tile's first column-tile row too, since tiles are on or below the diagonal)
        A += m0;
        B += m0;
        K -= m0;
    }
    // De-phase the two workgroups that share a CU.  All workgroups of a launch take the same time, so the
    // pair that starts together stays in lockstep: both wait on their C-tile loads, both fight for the
    // matrix pipe, both store.  The dispatcher fills every CU once before placing second workgroups, so
    // blocks 256..511 are the late partners of blocks 0..255 (observed; speed only): they sleep `stagger` x
    // 2048 cycles once, and every later workgroup inherits the offset of the slot it replaces.
    if (stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(32);       // 32 x 64 cycles
    }

    // The accumulators start as beta*C (all loads of the tile issued back to back, one wait) and the
    // sign rides on the staged A operand, so the epilogue is stores only.  (A load-modify-store epilogue
    // serialises 64 global round trips per thread: stores may alias the next load.)
    const int fr = lane & 15, fq = lane >> 4;
    gs_d4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = n0 + (wn * WN + j) * 16 + fr;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int row = m0 + (wm * WM + i) * 16 + fq + 4 * x;
                acc[i][j][x] = (beta && row < M && col < N) ? C[(int64_t)row * ldc + col] : 0.0;
            }
        }

    gs_d2 ra[PF][A_IT], rb[PF][B_IT];
    // one 16-B global load of the staging set: i < A_IT -> A tile, else B tile; `slot` = register set (compile-time)
    auto gload_one = [&](int kc, int i, auto slot) {
        constexpr int S = decltype(slot)::value;
        if (i < A_IT) {
            const int vv = t + i * NT;
            if (vv < A_VECS) {
                int row = m0 + (vv >> 3);
                row = row < M ? row : M - 1;
                ra[S][i] = *reinterpret_cast<const gs_d2*>(A + (int64_t)row * lda + kc * GS_KC + 2 * (vv & 7));
            }
        } else {
            const int vv = t + (i - A_IT) * NT;
            if (vv < B_VECS) {
                int row = n0 + (vv >> 3);
                row = row < N ? row : N - 1;
                rb[S][i - A_IT] = *reinterpret_cast<const gs_d2*>(B + (int64_t)row * ldb + kc * GS_KC + 2 * (vv & 7));
            }
        }
    };
    auto gload = [&](int kc, auto slot) {
#pragma unroll
        for (int i = 0; i < A_IT + B_IT; ++i) gload_one(kc, i, slot);
    };
    auto swrite = [&](int stage, auto slot) {
        constexpr int S = decltype(slot)::value;
        double* sA = lds + stage * (BM + BN) * GS_LSTR;
        double* sB = sA + BM * GS_LSTR;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int vv = t + it * NT;
            if (vv < A_VECS) {          // rows are only 8-B aligned at an odd stride: two 8-byte stores
                double* q = sA + (vv >> 3) * GS_LSTR + 2 * (vv & 7);
                q[0] = ra[S][it][0] * sign;
                q[1] = ra[S][it][1] * sign;
            }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int vv = t + it * NT;
            if (vv < B_VECS) {
                double* q = sB + (vv >> 3) * GS_LSTR + 2 * (vv & 7);
                q[0] = rb[S][it][0];
                q[1] = rb[S][it][1];
            }
        }
    };
    auto multiply = [&](int stage) {
        const double* sA = lds + stage * (BM + BN) * GS_LSTR + (wm * WM * 16 + fr) * GS_LSTR + fq;
        const double* sB = lds + stage * (BM + BN) * GS_LSTR + BM * GS_LSTR + (wn * WN * 16 + fr) * GS_LSTR + fq;
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
    };
    using I0 = std::integral_constant<int, 0>;

    const int nk = K / GS_KC;
    // STAMP build only (diagnostics, separate instantiation): shader-cycle sums of the loop phases
    unsigned long long ph[5] = {0, 0, 0, 0, 0}, tq = 0;
    auto stamp = [&](int i) {
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long now;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (i >= 0) ph[i] += now - tq;
            tq = now;
        }
    };
    if constexpr (PF == 1) {
        stamp(-1);
        gload(0, I0{});
        swrite(0, I0{});
        __syncthreads();
        stamp(0);                                   // prologue: C loads issued, first chunk staged
        for (int c = 0; c < nk; ++c) {
            // Next chunk's operands: issued in one burst ahead of the MFMAs.  (Spreading them over the k-steps
            // was measured and is no better: under load each load instruction blocks in-order issue for ~300
            // cycles wherever it sits; the CU's vector-memory path, ~7-10 B/clk, is the ceiling for this tile.)
            if (c + 1 < nk) gload(c + 1, I0{});
            stamp(1);                               // global load issue
            multiply(c & 1);
            stamp(2);                               // fragment reads + MFMAs
            if (c + 1 < nk) swrite((c + 1) & 1, I0{});
            stamp(3);                               // wait for the global loads, LDS stores
            __syncthreads();
            stamp(4);                               // barrier
        }
    } else {
        static_assert(PF == 1 || PF == 4, "ring of four register sets");
        using I1 = std::integral_constant<int, 1 % PF>;
        using I2 = std::integral_constant<int, 2 % PF>;
        using I3 = std::integral_constant<int, 3 % PF>;
        // nk is a multiple of 4 (the launcher checks K % 64 == 0).  Branch-free body: loads past the end re-read the last
        // chunk and the last LDS store goes to the stage nobody reads again -- a branch around a load makes the
        // compiler's wait-count bookkeeping drain every outstanding load at the join.
        auto clampk = [&](int kc) { return kc < nk ? kc : nk - 1; };
        gload(0, I0{});
        gload(clampk(1), I1{});
        gload(clampk(2), I2{});
        gload(clampk(3), I3{});
        swrite(0, I0{});
        __syncthreads();
        auto iter = [&](int c, auto slot, auto next) {
            gload(clampk(c + 4), slot);             // this set went to LDS in the previous iteration
            multiply(c & 1);
            swrite((c + 1) & 1, next);              // requested three iterations ago
            __syncthreads();
        };
        for (int c = 0; c < nk; c += 4) {
            iter(c, I0{}, I1{});
            iter(c + 1, I1{}, I2{});
            iter(c + 2, I2{}, I3{});
            iter(c + 3, I3{}, I0{});
        }
    }
    if (STAMP && stamps && lane == 0) {
        unsigned long long* o = stamps + ((int64_t)blockIdx.x * (NT / 64) + w) * 5;
        for (int i = 0; i < 5; ++i) o[i] = ph[i];
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
                if (row < M && col < N) C[(int64_t)row * ldc + col] = acc[i][j][x];
            }
        }
}

