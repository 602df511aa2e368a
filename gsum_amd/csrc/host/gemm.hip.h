// GEMM launcher: tile configurations, the bulk tile's dispatch
// (part of gsum_capi.hip: included from there, in order -- one translation unit)
#pragma once
// ---- GEMM launcher ----------------------------------------------------------------------------
template <int WM, int WN, int WAVES_M, int WAVES_N, int PF = 1>
static int gs_launch_gemm(gsum_ctx* ctx, hipStream_t s, double* C, int64_t ldc, const double* A, int64_t lda,
                          const double* B, int64_t ldb, int64_t M, int64_t N, int K, int tri, int beta, double sign) {
    constexpr int BM = WM * 16 * WAVES_M, BN = WN * 16 * WAVES_N;
    if (M <= 0 || N <= 0) return 0;
    if (K % GS_KC != 0) GS_FAIL("gemm: K must be a multiple of 16");
    const size_t shmem = 2 * (size_t)(BM + BN) * GS_LSTR * sizeof(double);
    if (PF > 1 && K % (GS_KC * PF) != 0) GS_FAIL("gemm: the prefetch ring needs K to be a multiple of 64");
    auto kern = k_gemm_nt<WM, WN, WAVES_M, WAVES_N, false, PF>;
    if (!ctx->lds_attr_done.count((const void*)kern)) {          // per context: the attribute is per device
        GS_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        ctx->lds_attr_done.insert((const void*)kern);
    }
    int64_t blocks;
    if (tri) {
        if (M != N || BM != BN) GS_FAIL("gemm: tri mode needs a square C and square tiles");
        int64_t T = (M + BM - 1) / BM;
        blocks = T * (T + 1) / 2;
    } else {
        blocks = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * WAVES_M * WAVES_N), shmem, s, C, ldc, A, lda, B, ldb, (int)M,
                       (int)N, K, tri, beta, sign, (unsigned long long*)nullptr,
                       (BM == 128 && BN == 128) ? std::min(K / 16, 32) : 0);
    GS_CHECK(hipGetLastError());
    return 0;
}

// cfg 1:  32x128 tile (1x4 waves of 32x32)   -- chain GEMMs: sibling-column / look-ahead updates, gradient and predict sweeps
// cfg 2:  16x256 tile (1x4 waves of 16x64)   -- border rows (RHS^T) only
// cfg 5: 128x128 tile (2x4 waves of 64x32, register staging) -- stand-in for the bulk tile when operands are not 16-B aligned
// cfg 7: 128x64 tile, LDS-direct operand staging, three workgroups per CU -- the bulk trailing update (k_gemm_ld3)
static int gs_dispatch(gsum_ctx* ctx, hipStream_t s, int cfg, double* C, int64_t ldc, const double* A, int64_t lda,
                       const double* B, int64_t ldb, int64_t M, int64_t N, int K, int tri, int beta, double sign) {
    // the LDS-direct loads fetch 16 B per lane: operands must be 16-B aligned with even leading dimensions (true for
    // every matrix this library allocates); anything else takes the register-staged tile, which gives the same bits
    if (cfg == 7 && (((uintptr_t)A | (uintptr_t)B) & 15 || (lda & 1) || (ldb & 1))) cfg = 5;
    if (ctx->first_tiles && cfg != 7) GS_FAIL("internal: only the cfg-7 bulk tile counts first-column tiles");
    if (cfg == 8 && (((uintptr_t)A | (uintptr_t)B) & 15 || (lda & 1) || (ldb & 1))) cfg = 5;
    if (cfg == 8) {                                   // 128 x 128 tiles, 64 x 32 wave tiles, two workgroups per CU (round 5: large trailing matrices)
        if (M <= 0 || N <= 0) return 0;
        if (K % GS_KC != 0) GS_FAIL("gemm: K must be a multiple of 16");
        if (tri == 2) GS_FAIL("gemm: the 128 x 128 tile has no tri = 2 form");
        const size_t shmem = 2 * (size_t)((128 + 128) * GS_KC + 4) * sizeof(double);
        const void* kfn = (const void*)k_gemm_ld3b;
        if (!ctx->lds_attr_done.count(kfn)) {
            GS_CHECK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
            ctx->lds_attr_done.insert(kfn);
        }
        int64_t blocks;
        if (tri) {
            if (M != N) GS_FAIL("gemm: tri mode needs a square C");
            const int64_t Tt = (M + 127) / 128;
            blocks = Tt * (Tt + 1) / 2;
        } else {
            blocks = ((M + 127) / 128) * ((N + 127) / 128);
        }
        hipLaunchKernelGGL(k_gemm_ld3b, dim3((unsigned)blocks), dim3(512), shmem, s, C, ldc, A, lda, B, ldb, (int)M, (int)N, K, tri, beta, sign,
                           ctx->kst_ptr);
        ctx->kst_ptr = nullptr;
        GS_CHECK(hipGetLastError());
        return 0;
    }
    if (cfg == 7) {                                   // 128 x 64 tiles, 32 x 32 wave tiles, 3 workgroups per CU: the bulk default
        if (M <= 0 || N <= 0) return 0;
        if (K % GS_KC != 0) GS_FAIL("gemm: K must be a multiple of 16");
        size_t shmem = 2 * (size_t)((128 + 64) * GS_KC + 4) * sizeof(double);
        // pad the request so that fewer bulk workgroups share a CU and chain kernels find LDS at once (look-ahead schedules)
        if (ctx->bulk_pad_now && ctx->bulk_lds_pad > 0) shmem = std::max(shmem, (size_t)ctx->bulk_lds_pad);
        const void* kfn = (const void*)k_gemm_ld3<2>;
        if (!ctx->lds_attr_done.count(kfn)) {
            GS_CHECK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
            ctx->lds_attr_done.insert(kfn);
        }
        int64_t blocks;
        if (tri) {
            if (M != N) GS_FAIL("gemm: tri mode needs a square C");
            const int64_t Tt = (M + 127) / 128;
            // (tri == 1: the partial row tile first, see gs_tri_tiles64 -- the kernel makes the same choice from the same values)
            blocks = tri == 1 ? gs_tri_tiles64(M) : Tt * (Tt + 1);
        } else {
            blocks = ((M + 127) / 128) * ((N + 63) / 64);
        }
        hipLaunchKernelGGL(k_gemm_ld3<2>, dim3((unsigned)blocks), dim3(512), shmem, s, C, ldc, A, lda, B, ldb, (int)M, (int)N, K, tri,
                           beta, sign, ctx->kst_ptr, tri == 2 ? 0 : ctx->first_tiles, ctx->first_done,
                           (tri == 1 && ctx->first_tiles > 0) ? ctx->second_c2 : 0, ctx->second_done);
        ctx->kst_ptr = nullptr;
        ctx->first_tiles = 0;
        ctx->first_done = nullptr;
        ctx->second_c2 = 0;
        ctx->second_done = nullptr;
        GS_CHECK(hipGetLastError());
        return 0;
    }
    switch (cfg) {
        case 1:
            if (ctx->chain_prefetch && K % 64 == 0) return gs_launch_gemm<2, 2, 1, 4, 4>(ctx, s, C, ldc, A, lda, B, ldb, M, N, K, tri, beta, sign);
            return gs_launch_gemm<2, 2, 1, 4>(ctx, s, C, ldc, A, lda, B, ldb, M, N, K, tri, beta, sign);
        case 2: return gs_launch_gemm<1, 4, 1, 4>(ctx, s, C, ldc, A, lda, B, ldb, M, N, K, tri, beta, sign);
        case 5: return gs_launch_gemm<4, 2, 2, 4>(ctx, s, C, ldc, A, lda, B, ldb, M, N, K, tri, beta, sign);
    }
    GS_FAIL("gemm: unknown tile configuration");
}

// cfg GS_BULK stands for the bulk trailing-update kernel (cfg 7); those launches are the ones the profile records as "bulk".
#define GS_BULK (-5)
static int gs_gemm(gsum_ctx* ctx, hipStream_t s, int cfg, double* C, int64_t ldc, const double* A, int64_t lda,
                   const double* B, int64_t ldb, int64_t M, int64_t N, int K, int tri, int beta, double sign) {
    const double algo_override = ctx->next_algo_flops;      // consumed by this call whether or not it is profiled
    ctx->next_algo_flops = -1.0;
    const bool bulk = cfg == GS_BULK;
    if (bulk) cfg = 7;
    if (M <= 0 || N <= 0) return 0;
    // algorithmic flops of the update: lower-triangular SYRK M(M+1)K, rectangular 2MNK
    double fl = tri ? (double)M * (double)(M + 1) * K : 2.0 * (double)M * (double)N * K;
    if (algo_override >= 0.0) fl = algo_override;                        // caller knows better (trapezoidal region)
    const int rec = gs_prof_begin(ctx, s, bulk ? GS_PROF_BULK : GS_PROF_PANEL, fl);
    const int rc = gs_dispatch(ctx, s, cfg, C, ldc, A, lda, B, ldb, M, N, K, tri, beta, sign);
    gs_prof_end(ctx, s, rec);
    return rc;
}

