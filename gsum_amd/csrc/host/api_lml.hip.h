// C ABI: gsum_lml_resident[_shard], gsum_lml_batch, gsum_shard_range
// (part of gsum_capi.hip: included from there, in order -- one translation unit)
#pragma once
static int gs_lml_on_sets(gsum_ctx* ctx, gs_inputs* I, const gsum_kernel_desc* kernels, int32_t n_kernels, double nugget,
                          double* G_out, double* sld_out, int64_t* info_out);

// set_of (or NULL): the right-hand-side set of every evaluation (gsum_set_inputs_sets)
static int gs_lml_on(gsum_ctx* ctx, gs_inputs* I, const gsum_kernel_desc* kernels, int32_t n_kernels, double nugget,
                     double* G_out, double* sld_out, int64_t* info_out, const int32_t* set_of = nullptr) {
    if (ctx && n_kernels == 0) return 0;                          // an empty batch is no work, whatever the pointers
    if (!ctx || !kernels || !G_out || !sld_out || !info_out || n_kernels < 0) return -2;
    if (set_of)
        for (int i = 0; i < n_kernels; ++i)
            if (set_of[i] < 0 || set_of[i] >= I->n_sets) GS_FAIL("set_of: no such right-hand-side set (gsum_set_inputs_sets)");
    ctx->set_of = set_of;
    const int rc = gs_lml_on_sets(ctx, I, kernels, n_kernels, nugget, G_out, sld_out, info_out);
    ctx->set_of = nullptr;
    return rc;
}

static int gs_lml_on_sets(gsum_ctx* ctx, gs_inputs* I, const gsum_kernel_desc* kernels, int32_t n_kernels, double nugget,
                          double* G_out, double* sld_out, int64_t* info_out) {
    GS_CHECK(hipSetDevice(ctx->device));
    ctx->in = I;
    if (!ctx->in->X) GS_FAIL("gsum_set_inputs has not been called");
    if (gs_check_order(ctx, ctx->in->n)) return -2;
    for (int i = 0; i < n_kernels; ++i)
        if (gs_check_desc(ctx, &kernels[i], ctx->in->d)) return -2;
    // (kernel trees take the one-workgroup-per-evaluation paths too since round 5: k_lml_small<true> / k_lml_medium<true> walk the
    // postfix program in their build step -- the reference's own workloads are 5 ... 20 points on 8000-point grids, models.py:958-960)
    if (ctx->in->n <= GS_NB && ctx->small_path) {
        ctx->cur = &ctx->slots[0];
        return gs_lml_small(ctx, kernels, n_kernels, nugget, G_out, sld_out, info_out);
    }
    // break-even against the grouped schedule, re-measured in round 4 (the grouped launches made small batches much faster than round 3's
    // 20 streams; tools/gpu_medium_breakeven.py, profiles/r04_medium_breakeven.log): the fused path wins from 2, ~24, ~56, ~104, ~130, ~190,
    // ~215 evaluations at n = 256, 512, 1024, 1536, 2048, 3072, 4096 -- n / 16 above n = 256 (round 3's rule n^1.55 / 2000 chose the fused
    // path up to 40 % too early: n = 2048, 96 evaluations 17.5 ms fused against 12.4 grouped)
    const int med_min = ctx->medium_min_batch > 0 ? ctx->medium_min_batch
                                                  : (ctx->in->n <= 256 ? 2 : std::max(4, (int)(ctx->in->n / 16)));
    if (ctx->in->n <= GS_MEDIUM_MAX && ctx->medium_path && n_kernels >= med_min) {
        ctx->cur = &ctx->slots[0];
        return gs_lml_medium(ctx, kernels, n_kernels, nugget, G_out, sld_out, info_out);
    }
    if (n_kernels >= ctx->wave_min) {
        ctx->cur = &ctx->slots[0];
        return gs_lml_wave(ctx, kernels, n_kernels, nugget, G_out, sld_out, info_out);
    }
    // one or two evaluations: one after the other, each with the schedule of a
    // single factorisation (look-ahead / persistent chain) on the context's own streams
    gs_slot* sl = &ctx->slots[0];
    ctx->cur = sl;
    ctx->batch_active = 1;
    int rc = 0;
    for (int i = 0; i < n_kernels && !rc; ++i) {
        rc = gs_eval_enqueue(ctx, &kernels[i], nugget, i);
        if (!rc) sl->pending = i;
        if (!rc) rc = gs_eval_harvest(ctx, sl, G_out, sld_out, info_out);
    }
    return rc;
}

int gsum_lml_resident(gsum_ctx* ctx, const gsum_kernel_desc* kernels, int32_t n_kernels, double nugget,
                      double* G_out, double* sld_out, int64_t* info_out) {
    if (!ctx) return -2;
    return gs_lml_on(ctx, &ctx->res, kernels, n_kernels, nugget, G_out, sld_out, info_out);
}

int gsum_lml_resident_sets(gsum_ctx* ctx, const gsum_kernel_desc* kernels, const int32_t* set_of, int32_t n_kernels, double nugget,
                           double* G_out, double* sld_out, int64_t* info_out) {
    if (!ctx) return -2;
    return gs_lml_on(ctx, &ctx->res, kernels, n_kernels, nugget, G_out, sld_out, info_out, set_of);
}

int gsum_shard_range(int64_t total, int32_t rank, int32_t world, int64_t* lo, int64_t* hi) {
    if (total < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return -2;
    const int64_t chunk = (total + world - 1) / world;
    *lo = std::min<int64_t>(total, (int64_t)rank * chunk);
    *hi = std::min<int64_t>(total, *lo + chunk);
    return 0;
}

int gsum_lml_resident_shard(gsum_ctx* ctx, const gsum_kernel_desc* kernels, int32_t n_kernels, int32_t rank, int32_t world,
                            double nugget, double* G_out, double* sld_out, int64_t* info_out, int64_t* lo, int64_t* hi) {
    if (!ctx) return -2;
    if (!kernels || !G_out || !sld_out || !info_out || !lo || !hi || n_kernels < 0) {
        ctx->err = "gsum_lml_resident_shard: null argument";
        return -2;
    }
    if (gsum_shard_range(n_kernels, rank, world, lo, hi)) {
        ctx->err = "gsum_lml_resident_shard: bad rank / world";
        return -2;
    }
    if (*hi == *lo) return 0;                       // more ranks than grid points: nothing for this one
    const int64_t kk = (int64_t)ctx->res.k * ctx->res.k;
    return gs_lml_on(ctx, &ctx->res, kernels + *lo, (int32_t)(*hi - *lo), nugget, G_out + *lo * kk, sld_out + *lo, info_out + *lo);
}

int gsum_lml_batch(gsum_ctx* ctx, const gsum_kernel_desc* kernels, int32_t n_kernels, const double* X, int64_t n,
                   int32_t d, const double* RHS, int32_t k, double nugget, double* G_out, double* sld_out,
                   int64_t* info_out) {
    if (!ctx) return -2;
    int rc = gs_upload_inputs(ctx, X, n, d, RHS, k);
    if (rc) return rc;
    return gs_lml_on(ctx, &ctx->op, kernels, n_kernels, nugget, G_out, sld_out, info_out);
}

