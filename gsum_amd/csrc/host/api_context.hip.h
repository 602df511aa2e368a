// C ABI: gsum_init / gsum_destroy, options
// (part of gsum_capi.hip: included from there, in order -- one translation unit)
#pragma once

static int gs_slot_init(gsum_ctx* ctx, gs_slot* sl) {
    // Slot 0 owns the context's streams (gsum_init: four streams on four command-processor pipes).  The gradient batch keeps up to
    // three more evaluations in flight, each entirely on ONE stream, while slot 0 uses only its main stream: they take slot 0's chain
    // and auxiliary streams and the third group's -- streams created later would share a pipe with one of these.
    const int idx = (int)(sl - ctx->slots);
    gs_slot* s0 = &ctx->slots[0];
    if (idx == 1 && s0->sp) { sl->sm = s0->sp; sl->own_sm = false; }
    else if (idx == 2 && s0->sa) { sl->sm = s0->sa; sl->own_sm = false; }
    else if (idx == 3 && ctx->wave.g[2].sc) { sl->sm = ctx->wave.g[2].sc; sl->own_sm = false; }
    else GS_CHECK(hipStreamCreateWithPriority(&sl->sm, hipStreamNonBlocking, ctx->prio_lo));
    GS_CHECK(hipEventCreateWithFlags(&sl->evFork, hipEventDisableTiming));
    for (int i = 0; i < 4; ++i) GS_CHECK(hipEventCreate(&sl->tev[i]));
    GS_CHECK(hipMalloc((void**)&sl->dres, 258 * sizeof(double)));
    GS_CHECK(hipMalloc((void**)&sl->dinfo, sizeof(int)));
    GS_CHECK(hipHostMalloc((void**)&sl->hres, 258 * sizeof(double), hipHostMallocDefault));
    return 0;
}

static int gs_need_slots(gsum_ctx* ctx, int n) {
    if (n > GS_MAX_SLOTS) n = GS_MAX_SLOTS;
    while (ctx->n_slots_ready < n) {
        if (gs_slot_init(ctx, &ctx->slots[ctx->n_slots_ready])) return -1;
        ++ctx->n_slots_ready;
    }
    return 0;
}

int gsum_init(int device, gsum_ctx** out) {
    if (!out) return -2;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_init_error = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0");
        return -1;
    }
    if (device < 0 || device >= count) {
        g_init_error = "device index out of range";
        return -2;
    }
    gsum_ctx* ctx = new gsum_ctx();
    ctx->device = device;
    auto fail = [&](const char* what, hipError_t err) {     // gsum_destroy: the streams, events and buffers created so far go with it
        g_init_error = std::string(what) + ": " + hipGetErrorString(err);
        gsum_destroy(ctx);
        return -1;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
    (void)hipDeviceGetStreamPriorityRange(&ctx->prio_lo, &ctx->prio_hi);   // hi = numerically lowest
    if (gs_need_slots(ctx, 1)) {
        g_init_error = ctx->err;
        ++ctx->n_slots_ready;               // the slot that failed half-way: gsum_destroy frees what it did create
        gsum_destroy(ctx);
        return -1;
    }
    ctx->cur = &ctx->slots[0];
    // Every stream the context's schedules run side by side is created HERE, back to back, before anything else touches the device:
    //   slot 0's main (low priority), chain and auxiliary streams (high)  -- the three parties of a single factorisation;
    //   one more high-priority stream                                     -- with the other two, the chain streams of a batch's three groups,
    //                                                                        whose bulk stream is slot 0's main stream.
    // The command processor serves a process' queues from FOUR pipes, assigned in the order the queues were created (index mod 4: every
    // order tried in round 4 fits, profiles/r04_stream_order.log): two streams that must run side by side on one pipe cost a batch
    // 3-6 % (325 -> 314 / 305 evals/s at n = 8192 for a chain-chain / chain-bulk pair) and a single factorisation 30-70 % (5.3 -> 7.0 /
    // 9.2 ms; with the round-3 probe, a 1-s time-out).  Four consecutive creations sit on four different pipes whatever the process
    // (torch, RCCL) created before.  A fourth group of a batch (option wave_groups = 4) creates a fifth stream and shares a pipe.
    if (gs_panel_stream(ctx, ctx->cur) || gs_aux_stream(ctx, ctx->cur)) {
        g_init_error = ctx->err;
        gsum_destroy(ctx);
        return -1;
    }
    if ((e = hipStreamCreateWithPriority(&ctx->wave.g[2].sc, hipStreamNonBlocking, ctx->prio_hi)) != hipSuccess) return fail("hipStreamCreateWithPriority", e);
    ctx->wave.g[2].own_sc = true;
    ctx->wave.sb = ctx->cur->sm;
    if ((e = hipMalloc((void**)&ctx->dstamps, 64 * sizeof(unsigned long long))) != hipSuccess) return fail("hipMalloc", e);
    (void)hipMemset(ctx->dstamps, 0, 64 * sizeof(unsigned long long));
    {   // Do the four streams really run side by side?  Probe every pair (gs_pipe_probe); the HIP runtime hands hardware queues out
        // in an order of its own (profiles/r05_pipe_probe.log: the first context of a process gets four queues of its own, the chain
        // and auxiliary streams of a SECOND context landed on one), so a high-priority stream that takes turns with another is
        // replaced by a newly created one -- which lands on another queue -- until all pairs overlap, eight tries at most.  The
        // rejects are destroyed only afterwards (a destroyed stream's queue would be the next one handed out).  Observable:
        // gsum_get_option "pipes_ok" / "pipe_overlap_permille" / "pipe_heals"; the binding warns once when it stays 0.
        std::vector<hipStream_t> rejects;
        for (int attempt = 0; attempt < 9; ++attempt) {
            hipStream_t* four[4] = {&ctx->cur->sm, &ctx->cur->sp, &ctx->cur->sa, &ctx->wave.g[2].sc};
            const hipStream_t now[4] = {*four[0], *four[1], *four[2], *four[3]};
            int ov[16];
            if (gs_pipe_probe(ctx, now, 4, ov)) {
                g_init_error = ctx->err;
                for (hipStream_t r : rejects) (void)hipStreamDestroy(r);
                gsum_destroy(ctx);
                return -1;
            }
            if (ctx->pipes_ok == 1 || attempt == 8) break;
            int victim = -1;                                    // the later stream of the first pair that took turns (never the main stream)
            for (int a = 0; a < 4 && victim < 0; ++a)
                for (int b = a + 1; b < 4 && victim < 0; ++b)
                    if (ov[a * 4 + b] < 500) victim = b;
            hipStream_t fresh = nullptr;
            if (victim < 1 || hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, ctx->prio_hi) != hipSuccess) break;
            rejects.push_back(*four[victim]);
            *four[victim] = fresh;
            ++ctx->pipe_heals;
        }
        for (hipStream_t r : rejects) (void)hipStreamDestroy(r);
        ctx->wave.sb = ctx->cur->sm;
    }
#ifdef GSUM_LAB
    // (lab build only: the product library reads no environment variable -- its ten options are set through gsum_set_option)
    const char* la = getenv("GSUM_LOOKAHEAD");
    if (la) ctx->lookahead = atoi(la);
    const char* pg = getenv("GSUM_PIVOT_GUARD_ULPS");
    if (pg) {
        const double g = (double)std::max(0, std::min(1024, atoi(pg))) * 2.220446049250313e-16;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(gs_pivot_guard), &g, sizeof g);
    }
    const char* cp = getenv("GSUM_CHAIN_PERSIST");
    if (cp) ctx->chain_persist = atoi(cp) < 0 ? -1 : (atoi(cp) != 0);
#endif
    *out = ctx;
    return 0;
}

void gsum_destroy(gsum_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < ctx->n_slots_ready; ++i) {
        gs_slot* sl = &ctx->slots[i];
        gs_mat_release(sl->ws);
        if (sl->dres) (void)hipFree(sl->dres);
        if (sl->dinfo) (void)hipFree(sl->dinfo);
        if (sl->hres) (void)hipHostFree(sl->hres);
        for (auto ev : sl->evP) (void)hipEventDestroy(ev);
        for (auto ev : sl->evM) (void)hipEventDestroy(ev);
        for (auto ev : sl->evA) (void)hipEventDestroy(ev);
        if (sl->evFork) (void)hipEventDestroy(sl->evFork);
        for (int k = 0; k < 4; ++k)
            if (sl->tev[k]) (void)hipEventDestroy(sl->tev[k]);
        if (sl->sm && sl->own_sm) (void)hipStreamDestroy(sl->sm);
        if (sl->sp) (void)hipStreamDestroy(sl->sp);
        if (sl->su && sl->own_su) (void)hipStreamDestroy(sl->su);
        if (sl->evU) (void)hipEventDestroy(sl->evU);
        if (sl->gws) (void)hipFree(sl->gws);
        if (sl->hgrad) (void)hipHostFree(sl->hgrad);
        if (sl->sa) (void)hipStreamDestroy(sl->sa);
        if (sl->evC) (void)hipEventDestroy(sl->evC);
        if (sl->evS) (void)hipEventDestroy(sl->evS);
        if (sl->evN) (void)hipEventDestroy(sl->evN);
    }
    gs_wave_release(ctx, true);
    for (gs_inputs* I : {&ctx->op, &ctx->res}) {
        if (I->X) (void)hipFree(I->X);
        if (I->Z) (void)hipFree(I->Z);
    }
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->panel_stats) (void)hipFree(ctx->panel_stats);
    if (ctx->hbatch) (void)hipHostFree(ctx->hbatch);
    if (ctx->gws) (void)hipFree(ctx->gws);
    if (ctx->dstamps) (void)hipFree(ctx->dstamps);
    for (auto ev : ctx->prof_pool) (void)hipEventDestroy(ev);
    delete ctx;
}

const char* gsum_last_error(gsum_ctx* ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

int64_t gsum_get_option(gsum_ctx* ctx, const char* name) {
    if (!ctx || !name) return -1;
    // ---- the contract (include/gsum_hip.h)
    if (!strcmp(name, "wave_streams")) return ctx->wave_last_streams;     // streams the last batch call used (groups + 1; 0: none yet)
    if (!strcmp(name, "wave_groups")) return ctx->wave_groups;
    if (!strcmp(name, "wave_size")) return ctx->wave_size;
    if (!strcmp(name, "lookahead")) return ctx->lookahead;
    if (!strcmp(name, "chain_persist")) return ctx->chain_persist;
    if (!strcmp(name, "chain_probe")) return ctx->chain_probe;          // 0 not run, 1 streams concurrent, -1 serialised
    if (!strcmp(name, "chain_aborts")) return ctx->chain_aborts;
    if (!strcmp(name, "uploads_skipped")) return ctx->uploads_skipped;
    if (!strcmp(name, "pipes_ok")) return ctx->pipes_ok;
    if (!strcmp(name, "pipe_overlap_permille")) return ctx->pipe_overlap_permille;
    if (!strcmp(name, "pipe_heals")) return ctx->pipe_heals;
    if (!strcmp(name, "profile_gemm")) return ctx->profile_gemm;
    if (!strcmp(name, "small_path")) return ctx->small_path;
    if (!strcmp(name, "medium_path")) return ctx->medium_path;
    if (!strcmp(name, "medium_min_batch")) return ctx->medium_min_batch;
#ifdef GSUM_LAB
    // ---- the lab (include/gsum_hip_debug.h)
    if (!strcmp(name, "batch_slots")) return ctx->batch_slots;
    if (!strcmp(name, "wave_depth")) return ctx->wave_depth;
    if (!strcmp(name, "wave_deep_rows")) return ctx->wave_deep_rows;
    if (!strcmp(name, "wave_near_on_chain")) return ctx->wave_near_on_chain;
    if (!strcmp(name, "wave_serial")) return ctx->wave_serial;
    if (!strcmp(name, "wave_shift")) return ctx->wave_shift;
    if (!strcmp(name, "wave_min")) return ctx->wave_min;
    if (!strcmp(name, "chain_rows")) return ctx->chain_rows;
    if (!strcmp(name, "chain_deep")) return ctx->chain_deep;
    if (!strcmp(name, "chain_depth")) return ctx->chain_depth;
    if (!strcmp(name, "lazy_far")) return ctx->lazy_far;
    if (!strcmp(name, "panel_wave_ticks") || !strcmp(name, "panel_waves")) {         // read-back of option panel_stats (synchronises)
        if (!ctx->panel_stats) return -1;
        unsigned long long h[2] = {0, 0};
        if (hipSetDevice(ctx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(h, ctx->panel_stats, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        return (int64_t)h[!strcmp(name, "panel_wave_ticks") ? 0 : 1];
    }
#endif
    return -1;
}

#ifdef GSUM_LAB
// the lab's switches (include/gsum_hip_debug.h): schedule variants and diagnostics, all bit-identical in results
static int gs_set_option_lab(gsum_ctx* ctx, const char* name, int64_t value) {
    if (!strcmp(name, "build_lower_only")) ctx->build_lower_only = (int)value;
    else if (!strcmp(name, "diag_stamps")) ctx->diag_stamps = (int)value;
    else if (!strcmp(name, "lazy_far")) ctx->lazy_far = (int)value;
    else if (!strcmp(name, "predict_lazy")) ctx->predict_lazy = value != 0;
    else if (!strcmp(name, "predict_panel256")) ctx->predict_panel256 = value != 0;
    else if (!strcmp(name, "predict_lookahead")) ctx->predict_lookahead = value != 0;
    else if (!strcmp(name, "predict_depth")) ctx->predict_depth = (int)std::max<int64_t>(1, std::min<int64_t>(8, value));
    else if (!strcmp(name, "predict_split")) ctx->predict_split = value != 0;
    else if (!strcmp(name, "medium_lazy")) {            // (process-wide: a __device__ variable of the code object)
        const int v = (int)std::max<int64_t>(1, std::min<int64_t>(64, value));      // grouping depth: 1 none, 2 pairs, ..., >= 16: left-looking at n <= 4096
        GS_CHECK(hipSetDevice(ctx->device));
        GS_CHECK(hipDeviceSynchronize());
        GS_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(gs_medium_lazy), &v, sizeof v));
    }
    else if (!strcmp(name, "lazy_min_np")) ctx->lazy_min_np = (int)std::max<int64_t>(1024, value);
    else if (!strcmp(name, "bench_fill")) ctx->bench_fill = (int)value;
    else if (!strcmp(name, "panel_stats")) {              // 1: (re)start accumulating wave lifetimes of k_panel256, 0: stop
        GS_CHECK(hipSetDevice(ctx->device));
        GS_CHECK(hipDeviceSynchronize());
        if (value && !ctx->panel_stats) GS_CHECK(hipMalloc((void**)&ctx->panel_stats, 2 * sizeof(unsigned long long)));
        if (value) GS_CHECK(hipMemset(ctx->panel_stats, 0, 2 * sizeof(unsigned long long)));
        if (!value && ctx->panel_stats) { (void)hipFree(ctx->panel_stats); ctx->panel_stats = nullptr; }
    }
    else if (!strcmp(name, "bulk_lds_pad")) ctx->bulk_lds_pad = (int)std::max<int64_t>(0, std::min<int64_t>(80 * 1024, value));
    else if (!strcmp(name, "chain_fused")) ctx->chain_fused = value < 0 ? -1 : (value != 0);
    else if (!strcmp(name, "la_depth2")) ctx->la_depth2 = value != 0;
    else if (!strcmp(name, "chain_prefetch")) ctx->chain_prefetch = value != 0;
    else if (!strcmp(name, "chain_min_np")) ctx->chain_min_np = (int)std::max<int64_t>(512, value);
    else if (!strcmp(name, "chain_rows")) ctx->chain_rows = value >= 512 ? 512 : 256;
    else if (!strcmp(name, "chain_lazy")) ctx->chain_lazy = value < 0 ? -1 : (int)std::min<int64_t>(2, value);
    else if (!strcmp(name, "chain_deep")) ctx->chain_deep = value < 0 ? -1 : (value != 0);
    else if (!strcmp(name, "chain_depth")) ctx->chain_depth = (int)std::max<int64_t>(2, std::min<int64_t>(8, value));
    else if (!strcmp(name, "chain_deep_rows")) ctx->chain_deep_rows = (int)std::max<int64_t>(0, value);
    else if (!strcmp(name, "chain_test_abort")) ctx->chain_test_abort = (int)std::max<int64_t>(0, value);
    else if (!strcmp(name, "chain_stamps")) ctx->chain_stamps = value != 0;
    else if (!strcmp(name, "batch_slots")) ctx->batch_slots = (int)std::max<int64_t>(1, std::min<int64_t>(GS_MAX_SLOTS, value));
    else if (!strcmp(name, "wave_shift")) ctx->wave_shift = (int)std::max<int64_t>(-1, value);
    else if (!strcmp(name, "wave_tile128_rows")) ctx->wave_tile128_rows = (int)std::max<int64_t>(0, value);
    else if (!strcmp(name, "wave_far_own")) ctx->wave_far_own = value != 0;
    else if (!strcmp(name, "wave_cohorts")) ctx->wave_cohorts = value >= 2 ? 2 : 1;
    else if (!strcmp(name, "wave_cohort_min")) ctx->wave_cohort_min = (int)std::max<int64_t>(1, value);
    else if (!strcmp(name, "wave_tail_rows")) ctx->wave_tail_rows = (int)value;
    else if (!strcmp(name, "wave_long_rounds")) ctx->wave_long_rounds = (int)std::max<int64_t>(2, value);
    else if (!strcmp(name, "wave_min")) ctx->wave_min = (int)std::max<int64_t>(1, value);
    else if (!strcmp(name, "grad_batch_wave")) ctx->grad_batch_wave = value != 0;
    else if (!strcmp(name, "grad_interleave")) ctx->grad_interleave = value != 0;
    else if (!strcmp(name, "grad_split")) ctx->grad_split = value != 0;
    else if (!strcmp(name, "grad_lazy_chain")) ctx->grad_lazy_chain = value != 0;
    else if (!strcmp(name, "wave_head")) ctx->wave_head = (int)std::max<int64_t>(0, value);
    else if (!strcmp(name, "wave_panel_rows_lds")) ctx->wave_panel_rows_lds = value != 0;
    else if (!strcmp(name, "wave_near_on_chain")) ctx->wave_near_on_chain = value != 0;
    else if (!strcmp(name, "wave_serial")) ctx->wave_serial = value != 0;
    else if (!strcmp(name, "wave_panel_wg4")) ctx->wave_panel_wg4 = value == 8 ? 8 : (value != 0 ? 4 : 0);
    else if (!strcmp(name, "wave_depth")) ctx->wave_depth = (int)std::max<int64_t>(1, std::min<int64_t>(8, value));
    else if (!strcmp(name, "wave_deep_rows")) ctx->wave_deep_rows = (int)std::max<int64_t>(0, value);
    else GS_FAIL(std::string("unknown option: ") + name);
    return 0;
}
#endif

int gsum_set_option(gsum_ctx* ctx, const char* name, int64_t value) {
    if (!ctx || !name) return -2;
    if (!strcmp(name, "lookahead")) ctx->lookahead = (int)value;
    else if (!strcmp(name, "profile_gemm")) {
        ctx->profile_gemm = (int)std::max<int64_t>(0, value);
        ctx->prof_eval_count = 0;
        ctx->prof_this_eval = true;
    }
    else if (!strcmp(name, "release_scratch")) {
        // hand the grown work buffers back (the medium path keeps up to 40 GB, the gradient path 2 n^2 doubles)
        GS_CHECK(hipSetDevice(ctx->device));
        GS_CHECK(hipDeviceSynchronize());
        if (ctx->scratch) GS_CHECK(hipFree(ctx->scratch));
        if (ctx->gws) GS_CHECK(hipFree(ctx->gws));
        ctx->scratch = ctx->gws = nullptr;
        ctx->scratch_cap = ctx->gws_cap = 0;
        gs_wave_release(ctx, false);                           // the groups' workspaces (their streams stay)
        for (int i = 0; i < ctx->n_slots_ready; ++i) {         // and the per-slot workspace matrices of the fused path
            gs_mat_release(ctx->slots[i].ws);
            ctx->slots[i].ws = nullptr;
            if (ctx->slots[i].gws) GS_CHECK(hipFree(ctx->slots[i].gws));      // ... and gradient buffers (3 n^2 doubles each)
            ctx->slots[i].gws = nullptr;
            ctx->slots[i].gws_cap = 0;
        }
    }
    else if (!strcmp(name, "small_path")) ctx->small_path = (int)value;
    else if (!strcmp(name, "medium_path")) ctx->medium_path = (int)value;
    else if (!strcmp(name, "medium_min_batch")) ctx->medium_min_batch = value > 0 ? (int)value : -1;
    else if (!strcmp(name, "chain_persist")) ctx->chain_persist = value < 0 ? -1 : (value != 0);
    else if (!strcmp(name, "pivot_guard_ulps")) {        // (process-wide: a __device__ variable of the code object)
        const double g = (double)std::max<int64_t>(0, std::min<int64_t>(1024, value)) * 2.220446049250313e-16;
        GS_CHECK(hipSetDevice(ctx->device));
        GS_CHECK(hipDeviceSynchronize());
        GS_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(gs_pivot_guard), &g, sizeof g));
    }
    else if (!strcmp(name, "wave_groups")) ctx->wave_groups = (int)std::max<int64_t>(1, std::min<int64_t>(GS_WV_STREAM_GROUPS, value));
    else if (!strcmp(name, "wave_size")) ctx->wave_size = (int)std::max<int64_t>(1, std::min<int64_t>(GS_WVC_MAX, value));
    else {
#ifdef GSUM_LAB
        return gs_set_option_lab(ctx, name, value);
#else
        GS_FAIL(std::string("unknown option: ") + name + " (schedule experiments and diagnostics live in libgsum_hip_lab.so)");
#endif
    }
    return 0;
}

