// C ABI: timers, kernel profile; lab build: stamps, probes, the tile microbenchmark
// (part of gsum_capi.hip: included from there, in order -- one translation unit)
#pragma once
int gsum_timers(gsum_ctx* ctx, double* ms, int32_t n) {
    if (!ctx || !ms) return -2;
    for (int i = 0; i < n && i < 4; ++i) ms[i] = ctx->timers[i];
    if (n > 4) {
        GS_CHECK(hipSetDevice(ctx->device));
        unsigned long long st[8] = {0};
        GS_CHECK(hipMemcpy(st, ctx->dstamps, sizeof st, hipMemcpyDeviceToHost));
        for (int i = 4; i < n && i < 9; ++i) ms[i] = (double)st[i - 4];
        if (n > 9) ms[9] = ctx->host_enqueue_ms;
    }
    return 0;
}

#ifdef GSUM_LAB
// The pairwise stream probe of gsum_init on the context's four streams plus `extra` (0..4) streams created now: out holds the
// (4 + extra)^2 overlaps in 1/1000 of the probe kernels' length.  A stream created after the first four shares a pipe with one of
// them: this is how the probe itself is validated (tests/test_gpu_round5.py) and how profiles/r05_pipe_probe.log was taken.
int gsum_debug_pipe_probe(gsum_ctx* ctx, int32_t extra, int32_t* out) {
    if (!ctx || !out || extra < 0 || extra > 4) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    hipStream_t all[8] = {ctx->slots[0].sm, ctx->slots[0].sp, ctx->slots[0].sa, ctx->wave.g[2].sc};
    const int n = 4 + extra;
    for (int i = 4; i < n; ++i) GS_CHECK(hipStreamCreateWithPriority(&all[i], hipStreamNonBlocking, ctx->prio_hi));
    for (int i = 0; i < n * n; ++i) out[i] = i / n == i % n ? 1000 : -1;
    const int keep_ok = ctx->pipes_ok, keep_pm = ctx->pipe_overlap_permille;
    const int rc = gs_pipe_probe(ctx, all, n, out);
    ctx->pipes_ok = keep_ok;
    ctx->pipe_overlap_permille = keep_pm;
    for (int i = 4; i < n; ++i) (void)hipStreamDestroy(all[i]);
    return rc;
}

// Realtime stamps (100 MHz ticks, relative to the first) of the last persistent-chain factorisation on slot 0's workspace
// (option "chain_stamps" = 1): GS_CH_STAMPS = 16 per outer step -- D role 0 step begins, 1 its diagonal block is up to date,
// 2 T0 set, 3 block row k + 1 up to date, 4 TL set, 5 sibling update done, 6 T1 set; P wave 0: 8 rows ready, 9 T0 seen,
// 10 sibling update done, 11 T1 seen, 12 published, 13 first update task starts, 14 done.  Returns the steps written.
int gsum_debug_chain_stamps(gsum_ctx* ctx, double* out, int32_t max_steps, int32_t* steps) {
    if (!ctx || !out || !steps) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    gsum_mat* m = ctx->slots[0].ws;
    *steps = 0;
    if (!m || !m->cstamps) return 0;
    const int Sall = m->T / 2, S = std::min<int>(Sall, max_steps);
    std::vector<unsigned long long> h((size_t)Sall * (GS_CH_STAMPS + GS_CH_KSTAMPS));
    GS_CHECK(hipDeviceSynchronize());
    GS_CHECK(hipMemcpy(h.data(), m->cstamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const unsigned long long t0 = h.empty() ? 0 : h[0];
    const int WOUT = GS_CH_STAMPS + GS_CH_KSTAMPS;
    for (int s = 0; s < S; ++s) {
        for (int i = 0; i < GS_CH_STAMPS; ++i) {
            const unsigned long long v = h[(size_t)s * GS_CH_STAMPS + i];
            out[(size_t)s * WOUT + i] = v ? (double)(long long)(v - t0) : -1.0;
        }
        for (int i = 0; i < GS_CH_KSTAMPS; ++i) {
            const unsigned long long v = h[(size_t)Sall * GS_CH_STAMPS + (size_t)s * GS_CH_KSTAMPS + i];
            out[(size_t)s * WOUT + GS_CH_STAMPS + i] = (v && v != ~0ull) ? (double)(long long)(v - t0) : -1.0;
        }
    }
    *steps = S;
    return 0;
}

int gsum_debug_diag_stamps(gsum_ctx* ctx, int64_t* out64) {
    if (!ctx || !out64) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    GS_CHECK(hipDeviceSynchronize());
    GS_CHECK(hipMemcpy(out64, ctx->dstamps, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

#endif  // GSUM_LAB

int gsum_kernel_profile(gsum_ctx* ctx, double* ms5, double* flops5, int64_t* launches5) {
    if (!ctx || !ms5 || !flops5 || !launches5) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    GS_CHECK(hipDeviceSynchronize());
    for (int c = 0; c < GS_PROF_CLASSES; ++c) {
        ms5[c] = flops5[c] = 0.0;
        launches5[c] = 0;
    }
    for (auto& r : ctx->prof_recs) {
        float ms = 0.f;
        GS_CHECK(hipEventElapsedTime(&ms, ctx->prof_pool[r.e0], ctx->prof_pool[r.e1]));
        ms5[r.cls] += ms;
        flops5[r.cls] += r.flops;
        launches5[r.cls] += 1;
    }
    ctx->prof_recs.clear();
    ctx->prof_next = 0;
    return 0;
}

#ifdef GSUM_LAB
int gsum_probe_mfma_f64(gsum_ctx* ctx, int32_t iters, int32_t waves_per_simd, int32_t n_acc, double* out3) {
    if (!ctx || !out3 || iters <= 0 || waves_per_simd < 1 || waves_per_simd > 8) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    const int blocks = 256 * waves_per_simd;      // 256-thread blocks: one wave per SIMD each
    const size_t ob = (size_t)blocks * 256 * sizeof(double), sb = (size_t)blocks * 4 * 2 * sizeof(unsigned long long);
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, ob + sb)) return -1;
    unsigned long long* dst = (unsigned long long*)((char*)ctx->scratch + ob);
    auto launch = [&](int its) -> int {
        switch (n_acc) {
            case 1: hipLaunchKernelGGL(k_probe_mfma<1>, dim3(blocks), dim3(256), 0, ctx->cur->sm, ctx->scratch, its, dst); break;
            case 2: hipLaunchKernelGGL(k_probe_mfma<2>, dim3(blocks), dim3(256), 0, ctx->cur->sm, ctx->scratch, its, dst); break;
            case 4: hipLaunchKernelGGL(k_probe_mfma<4>, dim3(blocks), dim3(256), 0, ctx->cur->sm, ctx->scratch, its, dst); break;
            case 8: hipLaunchKernelGGL(k_probe_mfma<8>, dim3(blocks), dim3(256), 0, ctx->cur->sm, ctx->scratch, its, dst); break;
            case 16: hipLaunchKernelGGL(k_probe_mfma<16>, dim3(blocks), dim3(256), 0, ctx->cur->sm, ctx->scratch, its, dst); break;
            default: return -2;
        }
        return 0;
    };
    if (launch(64)) GS_FAIL("n_acc must be 1, 2, 4, 8 or 16");      // warm-up
    GS_CHECK(hipGetLastError());
    GS_CHECK(hipEventRecord(ctx->cur->tev[0], ctx->cur->sm));
    launch(iters);
    GS_CHECK(hipGetLastError());
    GS_CHECK(hipEventRecord(ctx->cur->tev[1], ctx->cur->sm));
    std::vector<unsigned long long> st((size_t)blocks * 8);
    GS_CHECK(hipMemcpyAsync(st.data(), dst, sb, hipMemcpyDeviceToHost, ctx->cur->sm));
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    float ms = 0.f;
    GS_CHECK(hipEventElapsedTime(&ms, ctx->cur->tev[0], ctx->cur->tev[1]));
    const double n_mfma = (double)iters * 16.0;   // per wave
    const double flops = (double)blocks * 4.0 * n_mfma * 2048.0;
    double cyc = 0.0, rt = 0.0;
    for (size_t i = 0; i < st.size(); i += 2) {
        cyc += (double)st[i];
        rt += (double)st[i + 1];
    }
    out3[0] = flops / (ms * 1e-3) / 1e12;                       // TFLOP/s
    out3[1] = cyc / ((double)blocks * 4.0) / n_mfma;            // shader cycles per MFMA per wave
    out3[2] = rt > 0 ? cyc / rt * 0.1 : 0.0;                    // GHz (s_memrealtime ticks at 100 MHz)
    return 0;
}

int gsum_probe_hbm_write(gsum_ctx* ctx, int64_t bytes, double* gbps) {
    if (!ctx || !gbps || bytes < 4096) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, (size_t)bytes)) return -1;
    const int64_t nvec = bytes / 16;
    hipLaunchKernelGGL(k_probe_store, dim3(2048), dim3(256), 0, ctx->cur->sm, (gs_d2*)ctx->scratch, nvec);
    GS_CHECK(hipEventRecord(ctx->cur->tev[0], ctx->cur->sm));
    hipLaunchKernelGGL(k_probe_store, dim3(2048), dim3(256), 0, ctx->cur->sm, (gs_d2*)ctx->scratch, nvec);
    GS_CHECK(hipEventRecord(ctx->cur->tev[1], ctx->cur->sm));
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    float ms = 0.f;
    GS_CHECK(hipEventElapsedTime(&ms, ctx->cur->tev[0], ctx->cur->tev[1]));
    *gbps = (double)(nvec * 16) / (ms * 1e-3) / 1e9;
    return 0;
}

int gsum_bench_gemm_nt(gsum_ctx* ctx, int32_t cfg, int32_t tri, int64_t M, int64_t N, int64_t K, int64_t lda,
                       int32_t reps, double* out2) {
    if (!ctx || !out2 || M <= 0 || N <= 0 || K <= 0 || reps <= 0 || (cfg != 99 && lda < K)) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (cfg == 99) {
        // pure MFMA issue rate: M workgroups of N threads (N a multiple of 64, <= 512), K rounds of `lda` (4 or 8) independent MFMAs per wave
        if (N % 64 || N > 512 || (lda != 4 && lda != 8)) return -2;
        if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, (size_t)M * N * 8)) return -1;
        hipStream_t s = ctx->cur->sm;
        auto launch = [&]() {
            if (lda == 4) hipLaunchKernelGGL(k_mfma_peak<4>, dim3((unsigned)M), dim3((unsigned)N), 0, s, (double*)ctx->scratch, (int)K);
            else hipLaunchKernelGGL(k_mfma_peak<8>, dim3((unsigned)M), dim3((unsigned)N), 0, s, (double*)ctx->scratch, (int)K);
        };
        launch();
        GS_CHECK(hipEventRecord(ctx->cur->tev[0], s));
        for (int r = 0; r < reps; ++r) launch();
        GS_CHECK(hipEventRecord(ctx->cur->tev[1], s));
        GS_CHECK(hipStreamSynchronize(s));
        GS_CHECK(hipGetLastError());
        float ms = 0.f;
        GS_CHECK(hipEventElapsedTime(&ms, ctx->cur->tev[0], ctx->cur->tev[1]));
        const double fl = (double)M * (double)(N / 64) * (double)K * (double)lda * 2048.0;      // 16 x 16 x 4 x 2 flops per MFMA
        out2[0] = fl * reps / (ms * 1e-3) / 1e12;
        out2[1] = ms * 1e3 / reps;
        return 0;
    }
    const size_t cb = (size_t)M * N * 8, ab = (size_t)M * lda * 8, bb = (size_t)N * lda * 8;
    const size_t oa = (cb + 255) / 256 * 256, ob = oa + (ab + 255) / 256 * 256;
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, ob + bb)) return -1;
    char* base = (char*)ctx->scratch;
    double *dC = (double*)base, *dA = (double*)(base + oa), *dB = (double*)(base + ob);
    hipStream_t s = ctx->cur->sm;
    hipLaunchKernelGGL(k_fill_random, dim3(2048), dim3(256), 0, s, dC, (int64_t)(cb / 8), 1u);
    hipLaunchKernelGGL(k_fill_random, dim3(2048), dim3(256), 0, s, dA, (int64_t)(ab / 8), 2u);
    hipLaunchKernelGGL(k_fill_random, dim3(2048), dim3(256), 0, s, dB, (int64_t)(bb / 8), 3u);
    if (ctx->bench_fill == 1) {              // all-zero operands and C: what most of an RBF matrix's trailing update multiplies (power probe)
        GS_CHECK(hipMemsetAsync(dC, 0, cb, s));
        GS_CHECK(hipMemsetAsync(dA, 0, ab, s));
        GS_CHECK(hipMemsetAsync(dB, 0, bb, s));
    }
    const double* Bop = tri ? dA : dB;       // SYRK: both operands are the same panel
    if (gs_gemm(ctx, s, cfg, dC, N, dA, lda, Bop, lda, M, N, (int)K, tri, 1, -1.0)) return -1;   // warm-up
    GS_CHECK(hipEventRecord(ctx->cur->tev[0], s));
    for (int r = 0; r < reps; ++r)
        if (gs_gemm(ctx, s, cfg, dC, N, dA, lda, Bop, lda, M, N, (int)K, tri, 1, -1.0)) return -1;
    GS_CHECK(hipEventRecord(ctx->cur->tev[1], s));
    GS_CHECK(hipStreamSynchronize(s));
    float ms = 0.f;
    GS_CHECK(hipEventElapsedTime(&ms, ctx->cur->tev[0], ctx->cur->tev[1]));
    const double fl = tri ? (double)M * (double)(M + 1) * K : 2.0 * (double)M * (double)N * K;
    out2[0] = fl * reps / (ms * 1e-3) / 1e12;
    out2[1] = ms * 1e3 / reps;
    return 0;
}

int gsum_debug_gemm_nt(gsum_ctx* ctx, int32_t cfg, int32_t tri, double* C, const double* A, const double* B,
                       int64_t M, int64_t N, int64_t K, int32_t beta, double sign) {
    if (!ctx || !C || !A || !B) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    const size_t cb = (size_t)M * N * 8, ab = (size_t)M * K * 8, bb = (size_t)N * K * 8;
    const size_t oa = (cb + 255) / 256 * 256, ob = oa + (ab + 255) / 256 * 256;
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, ob + bb)) return -1;
    char* base = (char*)ctx->scratch;
    double *dC = (double*)base, *dA = (double*)(base + oa), *dB = (double*)(base + ob);
    GS_CHECK(hipMemcpyAsync(dC, C, cb, hipMemcpyHostToDevice, ctx->cur->sm));
    GS_CHECK(hipMemcpyAsync(dA, A, ab, hipMemcpyHostToDevice, ctx->cur->sm));
    GS_CHECK(hipMemcpyAsync(dB, B, bb, hipMemcpyHostToDevice, ctx->cur->sm));
    if (gs_gemm(ctx, ctx->cur->sm, cfg, dC, N, dA, K, dB, K, M, N, (int)K, tri, beta, sign)) return -1;
    GS_CHECK(hipMemcpyAsync(C, dC, cb, hipMemcpyDeviceToHost, ctx->cur->sm));
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    return 0;
}

#endif  // GSUM_LAB

