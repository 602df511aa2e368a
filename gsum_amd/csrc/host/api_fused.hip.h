// C ABI, fused path: resident inputs, one evaluation on a slot, whole evaluations in one workgroup (small / medium)
// (part of gsum_capi.hip: included from there, in order -- one translation unit)
#pragma once
int gsum_set_inputs_sets(gsum_ctx* ctx, const double* X, int64_t n, int32_t d, const double* RHS_sets, int32_t n_sets, int32_t k) {
    if (!ctx) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    ctx->cur = &ctx->slots[0];
    if (d < 1 || d > GSUM_MAX_D) GS_FAIL("input dimension must be 1..GSUM_MAX_D");
    if (gs_upload_X(ctx, &ctx->res, X, n, d)) return -1;
    if (gs_upload_Z(ctx, &ctx->res, RHS_sets, n, k, n_sets)) return -1;
    if (ctx->upload_pending) GS_CHECK(hipStreamSynchronize(ctx->cur->sm));        // (nothing was copied: the bytes were already there)
    ctx->upload_pending = false;
    return 0;
}

int gsum_set_inputs(gsum_ctx* ctx, const double* X, int64_t n, int32_t d, const double* RHS, int32_t k) {
    return gsum_set_inputs_sets(ctx, X, n, d, RHS, 1, k);
}

// right-hand sides of evaluation i of the current call (device pointer)
static const double* gs_z_of(const gsum_ctx* ctx, int i) {
    return ctx->in->Z + (ctx->set_of ? (size_t)ctx->set_of[i] * (size_t)ctx->in->n * (size_t)ctx->in->k : 0);
}

int gsum_resident_shape(gsum_ctx* ctx, int64_t* n, int32_t* d, int32_t* k) {
    if (!ctx || !n || !d || !k) return -2;
    *n = ctx->res.X ? ctx->res.n : 0;
    *d = ctx->res.X ? ctx->res.d : 0;
    *k = ctx->res.X ? ctx->res.k : 0;
    return 0;
}

// host inputs of gsum_lml_batch / gsum_lml_grad: uploaded into the operator-level set, never into the resident one
static int gs_upload_inputs(gsum_ctx* ctx, const double* X, int64_t n, int32_t d, const double* RHS, int32_t k) {
    GS_CHECK(hipSetDevice(ctx->device));
    ctx->cur = &ctx->slots[0];
    if (d < 1 || d > GSUM_MAX_D) GS_FAIL("input dimension must be 1..GSUM_MAX_D");
    if (gs_upload_X(ctx, &ctx->op, X, n, d)) return -1;
    if (gs_upload_Z(ctx, &ctx->op, RHS, n, k)) return -1;
    if (ctx->upload_pending) GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    ctx->upload_pending = false;
    return 0;
}

// enqueue one evaluation on the current slot (asynchronous: nothing waits on the host)
static int gs_eval_enqueue(gsum_ctx* ctx, const gsum_kernel_desc* desc, double nugget, int eval_index = 0) {
    gs_slot* sl = ctx->cur;
    const auto h0 = std::chrono::steady_clock::now();
    struct HostTimer {
        gsum_ctx* c; std::chrono::steady_clock::time_point t0;
        ~HostTimer() { c->host_enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
    } host_timer{ctx, h0};
    if (!sl->ws || sl->ws->n != ctx->in->n) {
        GS_CHECK(hipStreamSynchronize(sl->sm));
        gs_mat_release(sl->ws);
        sl->ws = nullptr;
        if (gs_mat_alloc(ctx, ctx->in->n, &sl->ws)) return -1;
    }
    gsum_mat* m = sl->ws;
    sl->last_desc = *desc;
    sl->last_nugget = nugget;
    if (ctx->profile_gemm > 0) ctx->prof_this_eval = (ctx->prof_eval_count++ % ctx->profile_gemm) == 0;
    GS_CHECK(hipEventRecord(sl->tev[0], sl->sm));
    if (gs_build_into(ctx, sl->sm, m, desc, ctx->in->X, ctx->in->d, nugget, ctx->build_lower_only)) return -1;
    sl->last_index = eval_index;
    if (gs_set_border(ctx, sl->sm, m, gs_z_of(ctx, eval_index), ctx->in->k)) return -1;
    GS_CHECK(hipEventRecord(sl->tev[1], sl->sm));
    if (gs_potrf(ctx, m)) return -1;
    GS_CHECK(hipEventRecord(sl->tev[2], sl->sm));
    if (gs_finalize(ctx, m)) return -1;
    GS_CHECK(hipEventRecord(sl->tev[3], sl->sm));
    m->factored = false;               // workspace: always rebuilt by the next evaluation
    return 0;
}

// wait for the evaluation pending on a slot and copy its results out
static int gs_eval_harvest(gsum_ctx* ctx, gs_slot* sl, double* G_out, double* sld_out, int64_t* info_out) {
    const int i = sl->pending, k = ctx->in->k;
    if (i < 0) return 0;
    GS_CHECK(hipStreamSynchronize(sl->sm));
    if ((int64_t)sl->hres[257] == GS_INFO_CHAIN_ABORT) {
        // the persistent chain timed out (its streams did not run side by side): once more on the host-enqueued schedule
        ctx->chain_persist = 0;
        ++ctx->chain_aborts;
        gs_slot* keep = ctx->cur;
        ctx->cur = sl;
        const gsum_kernel_desc d = sl->last_desc;
        const int rc = gs_eval_enqueue(ctx, &d, sl->last_nugget, sl->last_index);
        ctx->cur = keep;
        if (rc) return rc;
        GS_CHECK(hipStreamSynchronize(sl->sm));
    }
    for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) G_out[(size_t)i * k * k + a * k + b] = sl->hres[a * 16 + b];
    sld_out[i] = sl->hres[256];
    info_out[i] = (int64_t)sl->hres[257];
    float ms = 0.f;
    for (int s = 0; s < 3; ++s) {
        GS_CHECK(hipEventElapsedTime(&ms, sl->tev[s], sl->tev[s + 1]));
        ctx->timers[s] = ms;
    }
    GS_CHECK(hipEventElapsedTime(&ms, sl->tev[0], sl->tev[3]));
    ctx->timers[3] = ms;
    sl->pending = -1;
    return 0;
}

// n <= 128: one fused workgroup per evaluation (k_lml_small), up to 512 evaluations per launch
static int gs_reserve_pinned(gsum_ctx* ctx, size_t bytes) {
    if (ctx->hbatch_cap >= bytes) return 0;
    if (ctx->hbatch) (void)hipHostFree(ctx->hbatch);
    ctx->hbatch = nullptr;
    ctx->hbatch_cap = 0;
    GS_CHECK(hipHostMalloc((void**)&ctx->hbatch, bytes, hipHostMallocDefault));
    ctx->hbatch_cap = bytes;
    return 0;
}

static int gs_lml_small(gsum_ctx* ctx, const gsum_kernel_desc* kernels, int n_kernels, double nugget, double* G_out,
                        double* sld_out, int64_t* info_out) {
    // Evaluations per launch: up to 4096 (eight rounds of the 512 resident workgroups; 256 KB of scratch each).  With 512 per
    // launch, a synchronisation, a pageable read-back and the host-side unpacking sat between every two rounds of a kernel
    // that runs ~0.2 ms per round.
    const int k = ctx->in->k, CH = std::min(4096, (n_kernels + 511) / 512 * 512);
    hipStream_t s = ctx->cur->sm;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t o_desc = 0, o_res = up((size_t)CH * sizeof(gsum_kernel_desc)), o_set = o_res + up((size_t)CH * 258 * 8),
                 o_scr = o_set + up((size_t)CH * sizeof(int32_t));
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, o_scr + (size_t)CH * GS_SMALL_SCRATCH * 8)) return -1;
    char* base = (char*)ctx->scratch;
    if (gs_reserve_pinned(ctx, (size_t)CH * 258 * 8)) return -1;
    double* hres = ctx->hbatch;
    const int32_t* dset = ctx->set_of ? (const int32_t*)(base + o_set) : nullptr;
    for (int lo = 0; lo < n_kernels; lo += CH) {
        const int cnt = std::min(CH, n_kernels - lo);
        GS_CHECK(hipMemcpyAsync(base + o_desc, kernels + lo, (size_t)cnt * sizeof(gsum_kernel_desc), hipMemcpyHostToDevice, s));
        if (dset) GS_CHECK(hipMemcpyAsync(base + o_set, ctx->set_of + lo, (size_t)cnt * sizeof(int32_t), hipMemcpyHostToDevice, s));
        bool tree = false;                        // a tree among this launch's descriptors: the instantiation that can walk one
        for (int e = 0; e < cnt; ++e) tree = tree || kernels[lo + e].n_ops > 0;
        hipLaunchKernelGGL(tree ? k_lml_small<true> : k_lml_small<false>, dim3(cnt), dim3(256), 0, s, ctx->in->X, (int)ctx->in->n,
                           ctx->in->d, ctx->in->Z, k, (const gsum_kernel_desc*)(base + o_desc), nugget, (double*)(base + o_scr),
                           (double*)(base + o_res), dset);
        GS_CHECK(hipGetLastError());
        GS_CHECK(hipMemcpyAsync(hres, base + o_res, (size_t)cnt * 258 * 8, hipMemcpyDeviceToHost, s));
        GS_CHECK(hipStreamSynchronize(s));
        for (int e = 0; e < cnt; ++e) {
            const double* r = hres + (size_t)e * 258;
            for (int a = 0; a < k; ++a)
                for (int b = 0; b < k; ++b) G_out[(size_t)(lo + e) * k * k + a * k + b] = r[a * 16 + b];
            sld_out[lo + e] = r[256];
            info_out[lo + e] = (int64_t)r[257];
        }
    }
    return 0;
}

// 128 < n <= GS_MEDIUM_MAX (4096) with many evaluations: one workgroup per evaluation (k_lml_medium), 256 in flight
static int gs_lml_medium(gsum_ctx* ctx, const gsum_kernel_desc* kernels, int n_kernels, double nugget, double* G_out,
                         double* sld_out, int64_t* info_out) {
    const int k = ctx->in->k;
    const int64_t n = ctx->in->n, np = (n + GS_NB - 1) / GS_NB * GS_NB, T = np / GS_NB, ld = GS_LD(np);
    hipStream_t s = ctx->cur->sm;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const int64_t stride = (int64_t)(up((size_t)(np * ld + T * GS_NB * GS_NB + np + 16 * np) * 8) / 8);
    // evaluations per launch: whole rounds of the 512 resident workgroups (two per CU; a partial round would idle most
    // of the chip), within the memory budget below (512 x 134 MB at n = 4096 when the GPU is otherwise empty)
    size_t free_b = 0, total_b = 0;
    GS_CHECK(hipMemGetInfo(&free_b, &total_b));
    // what this call may hold: 80 % of what is free now plus the scratch it already owns, 80 GB at most
    const double budget = std::min(80e9, 0.8 * (double)free_b + (double)ctx->scratch_cap);
    const int64_t fit = (int64_t)(budget / (double)(stride * 8));
    const int cap = fit >= 512 ? 512 : (fit >= 256 ? 256 : (int)std::max<int64_t>(1, fit));
    const int CH = std::min(n_kernels, cap);
    const size_t o_desc = 0, o_res = up((size_t)CH * sizeof(gsum_kernel_desc)), o_set = o_res + up((size_t)CH * 258 * 8),
                 o_scr = o_set + up((size_t)CH * sizeof(int32_t));
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, o_scr + (size_t)CH * stride * 8)) return -1;
    char* base = (char*)ctx->scratch;
    const size_t shmem = (size_t)std::max<int>(GS_TILE_LD_DOUBLES, GS_DIAG_WS) * sizeof(double);
    for (const void* fn : {(const void*)k_lml_medium<false>, (const void*)k_lml_medium<true>})
        if (!ctx->lds_attr_done.count(fn)) {
            GS_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            ctx->lds_attr_done.insert(fn);
        }
    if (gs_reserve_pinned(ctx, (size_t)CH * 258 * 8)) return -1;
    double* hres = ctx->hbatch;
    const int32_t* dset = ctx->set_of ? (const int32_t*)(base + o_set) : nullptr;
    for (int lo = 0; lo < n_kernels; lo += CH) {
        const int cnt = std::min(CH, n_kernels - lo);
        GS_CHECK(hipMemcpyAsync(base + o_desc, kernels + lo, (size_t)cnt * sizeof(gsum_kernel_desc), hipMemcpyHostToDevice, s));
        if (dset) GS_CHECK(hipMemcpyAsync(base + o_set, ctx->set_of + lo, (size_t)cnt * sizeof(int32_t), hipMemcpyHostToDevice, s));
        bool tree = false;
        for (int e = 0; e < cnt; ++e) tree = tree || kernels[lo + e].n_ops > 0;
        hipLaunchKernelGGL(tree ? k_lml_medium<true> : k_lml_medium<false>, dim3(cnt), dim3(256), shmem, s, ctx->in->X, (int)n, ctx->in->d, ctx->in->Z, k,
                           (const gsum_kernel_desc*)(base + o_desc), nugget, (double*)(base + o_scr), stride, (double*)(base + o_res),
                           ctx->diag_stamps ? ctx->dstamps : (unsigned long long*)nullptr, dset);
        GS_CHECK(hipGetLastError());
        GS_CHECK(hipMemcpyAsync(hres, base + o_res, (size_t)cnt * 258 * 8, hipMemcpyDeviceToHost, s));
        GS_CHECK(hipStreamSynchronize(s));
        for (int e = 0; e < cnt; ++e) {
            const double* r = hres + (size_t)e * 258;
            for (int a = 0; a < k; ++a)
                for (int b = 0; b < k; ++b) G_out[(size_t)(lo + e) * k * k + a * k + b] = r[a * 16 + b];
            sld_out[lo + e] = r[256];
            info_out[lo + e] = (int64_t)r[257];
        }
    }
    return 0;
}

