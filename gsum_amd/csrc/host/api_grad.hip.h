// C ABI: value + gradient pieces (gsum_lml_grad[_batch])
// (part of gsum_capi.hip: included from there, in order -- one translation unit)
#pragma once
// Gradient pieces on top of one fused evaluation (see include/gsum_hip.h).  After the factorisation the workspace
// holds L, the 128x128 inverses of its diagonal blocks and W^T = RHS^T L^-T in the border rows; then
//   U = L^-T      right-looking sweep over block columns on an identity (rows below the current block are still
//                 zero and are skipped: n^3 / 3 flops, the GEMMs of the prediction path)
//   R^-1 = U U^T  one lower-tile SYRK launch whose tiles start their K loop at their own first row (n^3 / 3 flops)
//   V^T = W^T U^T (16 x n), then the fused kernel-gradient contractions, one grid row per hyperparameter.
static int gs_grad_check(gsum_ctx* ctx, const gsum_grad_param* params, int32_t n_params, int32_t d, int32_t k) {
    if (n_params < 1 || n_params > GSUM_MAX_GRAD) GS_FAIL("n_params must be 1..GSUM_MAX_GRAD");
    if (k < 1 || k > GSUM_MAX_RHS) GS_FAIL("k must be 1..GSUM_MAX_RHS");
    for (int p = 0; p < n_params; ++p) {
        const int code = params[p].code, dim = params[p].dim;
        if (code >= GSUM_GRAD_TREE_CONST && code <= GSUM_GRAD_TREE_ALPHA) {          // parameters of a kernel tree
            if (code <= GSUM_GRAD_TREE_WHITE ? (dim < 0 || dim >= GSUM_MAX_OPS) : (dim < 0 || (dim >> 4) >= GSUM_MAX_LEAVES || (dim & 15) >= d))
                GS_FAIL("gradient parameter of a kernel tree: slot / leaf / dimension out of range");
            continue;
        }
        if (code < GSUM_GRAD_AMPLITUDE || code > GSUM_GRAD_ADDITIVE) GS_FAIL("unknown gradient parameter code");
        if (code == GSUM_GRAD_LENGTH_DIM && (dim < 0 || dim >= d)) GS_FAIL("gradient parameter dim out of range");
    }
    return 0;
}

// One evaluation with gradient pieces, enqueued on slot `sl` (ctx->cur); results land in the slot's pinned buffers (hres: the
// fused evaluation's 258 doubles, hgrad: P x 257) when its main stream has drained.
//   solo: the single-evaluation schedule -- the U = L^-T sweep trails the look-ahead factorisation panel by panel on a stream
//         of its own, V^T runs beside the SYRK on the panel stream;
//  !solo: everything in order on the slot's main stream (a batch hides latencies with its other evaluations: gs_lml_on's rule).
// The stage is cut in three so that the sweep's launches can be ENQUEUED between the factorisation's (round 5, gs_potrf_chain's step hook):
// with the whole factorisation enqueued first -- ~280 launches, 3-4 ms of host time at n = 8192 -- the sweep's first launch reached its stream
// when the factorisation was almost over and "trailing" trailed nothing (kernel trace: k_set_identity at 5.1 ms of a 5.4-ms factorisation).
//   gs_grad_reserve   buffers, streams, events: everything that may allocate (a hipFree inside the hook would wait for a chain kernel that
//                     waits for launches the host has not enqueued yet);
//   gs_grad_prepare   the identity and the sweep stream's first waits;
//   gs_grad_sweep     block-column pairs [run->next_c, c_end) of U = L^-T;
//   gs_grad_post      whatever of the sweep is left, V^T, R^-1 = U U^T, the contractions, the read-back.

struct gs_grad_layout { size_t o_u, o_r, o_v, o_q, o_t, o_o, o_p, o_d, total; int chunks, rows_per; };
// split: room for the P stored lower triangles of dR (one evaluation alone: gs_grad_post)
static gs_grad_layout gs_grad_offsets(int64_t n, int64_t np, int P, bool split) {
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const int64_t ldg = np + GS_BORDER;
    gs_grad_layout L;
    L.chunks = (int)std::min<int64_t>(128, (n + 63) / 64);
    L.rows_per = (int)((n + L.chunks - 1) / L.chunks);
    L.o_u = 0;
    L.o_r = up((size_t)np * ldg * 8);
    L.o_v = L.o_r + up((size_t)np * ldg * 8);
    L.o_q = L.o_v + up((size_t)16 * ldg * 8);
    L.o_t = L.o_q + up((size_t)P * n * 16 * 8);
    L.o_o = L.o_t + up((size_t)P * n * 8);
    L.o_p = L.o_o + up((size_t)P * 257 * 8);
    L.o_d = L.o_p + up((size_t)P * L.chunks * 257 * 8);
    L.total = L.o_d + (split ? (size_t)P * up((size_t)np * ldg * 8) : 0);
    return L;
}

static int gs_grad_reserve(gsum_ctx* ctx, gs_slot* sl, int64_t n, int64_t np, int P, bool solo) {
    if (gs_reserve(ctx, &sl->gws, &sl->gws_cap, gs_grad_offsets(n, np, P, solo && ctx->grad_split).total)) return -1;
    if (!sl->hgrad) GS_CHECK(hipHostMalloc((void**)&sl->hgrad, (size_t)GSUM_MAX_GRAD * 257 * sizeof(double), hipHostMallocDefault));
    if (solo && !sl->su) {
        // the sweep runs beside the factorisation's main and panel streams: it takes the context's fourth stream (the third group's chain
        // stream of a batch, idle here) -- a stream created now would share a command-processor pipe with one of those two
        // (round 4 found the single gradient evaluation at 19.7 ms instead of 14.2 that way)
        if (sl == &ctx->slots[0] && ctx->wave.g[2].sc) { sl->su = ctx->wave.g[2].sc; sl->own_su = false; }
        else GS_CHECK(hipStreamCreateWithPriority(&sl->su, hipStreamNonBlocking, ctx->prio_lo));
        GS_CHECK(hipEventCreateWithFlags(&sl->evU, hipEventDisableTiming));
    }
    return 0;
}

static int gs_grad_prepare(gsum_ctx* ctx, gs_grad_run* r, gs_slot* sl, gsum_mat* m, int P, bool solo, bool on_chain) {
    *r = gs_grad_run();
    r->sl = sl; r->m = m; r->P = P; r->solo = solo; r->on_chain = on_chain;
    r->n = ctx->in->n; r->d = ctx->in->d;
    r->np = m->np; r->ld = m->ld; r->ldg = m->np + GS_BORDER;
    r->split = solo && ctx->grad_split;
    const gs_grad_layout L = gs_grad_offsets(r->n, r->np, P, r->split);
    if (sl->gws_cap < L.total || !sl->hgrad || (solo && !sl->su)) GS_FAIL("internal: gradient buffers were not reserved for this order");
    r->chunks = L.chunks; r->rows_per = L.rows_per;
    char* base = (char*)sl->gws;
    r->U = (double*)(base + L.o_u); r->Ri = (double*)(base + L.o_r); r->Vt = (double*)(base + L.o_v); r->Q = (double*)(base + L.o_q);
    r->trow = (double*)(base + L.o_t); r->dout = (double*)(base + L.o_o); r->part = (double*)(base + L.o_p);
    r->dR = (double*)(base + L.o_d);
    r->dr_stride = (int64_t)((((size_t)r->np * r->ldg * 8 + 255) / 256 * 256) / 8);
    r->s = sl->sm;
    // U = L^-T.  Solo: on a stream of its own, trailing the factorisation: block columns c, c + 1 of the sweep need the factor's
    // columns c0 .. c0 + 255 and their tables, which are final once the panel chain of that outer step has run (event
    // evP[c] of the look-ahead schedule, flag RP[s] of the persistent chain).  One factorisation alone is bound by its panel chain, with most
    // of the chip idle behind it -- the sweep's GEMMs (n^3 / 3 flops) fill that time instead of following it.
    r->trail = !on_chain && solo && ctx->lookahead != 0 && ctx->batch_active < 3;       // the condition under which gs_potrf records evP
    r->su = r->s;
    if (solo) {
        r->su = sl->su;
        GS_CHECK(hipEventRecord(sl->evU, r->s));                                  // everything enqueued so far (nothing of U is in use)
        if (!r->trail && !on_chain) GS_CHECK(hipStreamWaitEvent(r->su, sl->evU, 0));  // no per-panel events: the sweep follows the factorisation
        if (on_chain) GS_CHECK(hipStreamWaitEvent(r->su, sl->evFork, 0));         // (the chain's flags are zeroed on the main stream in front of that event)
    }
    hipLaunchKernelGGL(k_set_identity, dim3((unsigned)((r->np + 255) / 256), (unsigned)r->np), dim3(256), 0, r->su, r->U, r->ldg, (int)r->np);
    GS_CHECK(hipGetLastError());
    // two block columns per trailing update (K = 256), like the factorisation: halves the traffic of U's trailing part
    r->have_sib = m->Lsib != nullptr && (on_chain || (!solo && m->have_lsib));      // (the chain publishes the images pair by pair, ahead of RP[s])
    r->lazy_ok = ctx->predict_lazy && r->np >= ctx->lazy_min_np && !r->trail && (!on_chain || (ctx->grad_lazy_chain && r->np >= 10240));      // (a sweep that trails by EVENTS keeps the per-pair rhythm of the factorisation)
    r->prepared = true;
    return 0;
}

// block-column pairs [r->next_c, c_end) of the sweep; on_chain: a pair whose outer step is the factorisation's last waits for the chain
// kernel's end (evC), so it can only be enqueued behind gs_potrf_chain's last line
static int gs_grad_sweep(gsum_ctx* ctx, gs_grad_run* r, int c_end) {
    gsum_mat* m = r->m;
    gs_slot* sl = r->sl;
    const int64_t np = r->np, ld = r->ld, ldg = r->ldg;
    double* U = r->U;
    hipStream_t su = r->su;
    for (int c = r->next_c; c < c_end && c < m->T; c += 2) {
        const bool two = c + 1 < m->T;
        const int64_t c0 = (int64_t)c * GS_NB, c1 = c0 + GS_NB, r2 = two ? c1 + GS_NB : c1;
        if (r->trail) GS_CHECK(hipStreamWaitEvent(su, sl->evP[c], 0));
        if (r->on_chain) {
            const int S = m->T / 2, so = c / 2;
            if (so + 1 < S) {                 // a one-wave kernel that polls the flag in stream order (k_wait_flag), like the chain's own launches
                hipLaunchKernelGGL(k_wait_flag, dim3(1), dim3(64), 0, su, (const unsigned*)(m->cflags + gs_fl(GS_FL_RP, S, so)), 1u,
                                   (const unsigned*)nullptr, 0u, m->cflags);
                GS_CHECK(hipGetLastError());
            } else {
                GS_CHECK(hipStreamWaitEvent(su, sl->evC, 0));                  // the last outer step has no flag of this kind: the chain kernel's end
            }
        }
        // The pair's two panel solves and the sibling update between them: ONE k_panel256 launch where the factor's sibling images exist
        // as the sweep reaches the pair (the persistent chain publishes them step by step, the grouped batch leaves them in the pool) --
        // rows [c1, r2) of block column c are still zero and come out zero; otherwise k_panel / K = 128 GEMM / k_panel as in rounds 1-4.
        if (two && r->have_sib && ctx->predict_panel256) {
            if (gs_panel256(ctx, su, m, c, U + c0, ldg, r2)) return -1;
        } else {
            if (gs_trsm_rows(ctx, su, m, c, U + c0, ldg, c1)) return -1;
            if (two) {
                // rows below c1 are still zero in block column c: only rows < c1 feed the sibling column
                if (gs_gemm(ctx, su, 1, U + c1, ldg, U + c0, ldg, m->A + c1 * ld + c0, ld, c1, GS_NB, GS_NB, 0, 1, -1.0)) return -1;
                if (gs_trsm_rows(ctx, su, m, c + 1, U + c1, ldg, r2)) return -1;
            }
        }
        r->next_c = c + 2;
        if (r2 >= np) continue;
        // the trailing update, paired like the predictive sweep's (predict_lazy): after an even pair only the next pair's 256 columns take
        // this pair's update (K = 256), the pair after it applies both to everything right of it in one K = 512 launch -- rows [r2 - 256, r2)
        // of the older pair's columns are structural zeros.  Same products in the same ascending order per element.
        const bool pair = r->lazy_ok && two && c + 3 < m->T && r2 + 2 * GS_NB <= np;
        if (!r->deferred && pair) {
            if (gs_gemm(ctx, su, GS_BULK, U + r2, ldg, U + c0, ldg, m->A + r2 * ld + c0, ld, r2, 2 * GS_NB, (int)(r2 - c0), 0, 1, -1.0)) return -1;
            r->deferred = true;
        } else if (r->deferred) {
            const int64_t cp = c0 - 2 * GS_NB;
            if (gs_gemm(ctx, su, GS_BULK, U + r2, ldg, U + cp, ldg, m->A + r2 * ld + cp, ld, r2, np - r2, (int)(r2 - cp), 0, 1, -1.0)) return -1;
            r->deferred = false;
        } else if (gs_gemm(ctx, su, GS_BULK, U + r2, ldg, U + c0, ldg, m->A + r2 * ld + c0, ld, r2, np - r2, (int)(r2 - c0), 0, 1, -1.0))
            return -1;
    }
    return 0;
}

static int gs_grad_post(gsum_ctx* ctx, gs_slot* sl, gsum_mat* m, const gsum_kernel_desc* desc, const gsum_grad_param* params, int P, bool solo,
                        bool on_chain, gs_grad_run* started);

// gs_potrf_chain's step hook (ctx->chain_step_hook): outer step s of the factorisation is enqueued -- enqueue the sweep's pair of that step behind
// its RP[s] wait, so that it sits in its stream's queue when the flag comes
static int gs_grad_chain_hook(gsum_ctx* ctx, gsum_mat* m, int s) {
    gs_grad_run* r = &ctx->grad_run;
    gs_slot* sl = ctx->cur;
    if (!r->prepared && gs_grad_prepare(ctx, r, sl, m, ctx->chain_hook_P, true, true)) return -1;
    if (s + 1 >= m->T / 2) return 0;                    // (the last outer step's pair waits for evC: gs_grad_post)
    return gs_grad_sweep(ctx, r, 2 * (s + 1));
}

static int gs_grad_enqueue(gsum_ctx* ctx, gs_slot* sl, const gsum_kernel_desc* desc, const gsum_grad_param* params, int P, double nugget,
                           bool solo) {
    ctx->cur = sl;
    // everything the stage may allocate, before anything of the evaluation is enqueued (np: the larger of the two paddings a factorisation may take)
    {
        const int64_t n = ctx->in->n;
        const int64_t npg = std::max<int64_t>(gs_padded_order(ctx, n), (n + 2 * GS_NB - 1) / (2 * GS_NB) * (2 * GS_NB));
        if (gs_grad_reserve(ctx, sl, n, npg, P, solo)) return -1;
    }
    // the sweep below trails the factorisation: by the evP events of the host-enqueued schedule, or -- round 4 -- by the persistent
    // chain's own flags (RP[s]: the panel of outer step s is complete in every row), so that a gradient evaluation's factorisation
    // runs on the schedule a value-only evaluation gets (5.3 instead of 6.6 ms at n = 8192)
    ctx->chain_events_needed = 2;
    ctx->grad_run = gs_grad_run();
    if (solo && ctx->grad_interleave) {
        ctx->chain_hook_P = P;
        ctx->chain_step_hook = gs_grad_chain_hook;
    }
    const int rc_eval = gs_eval_enqueue(ctx, desc, nugget);
    ctx->chain_step_hook = nullptr;
    ctx->chain_events_needed = 0;
    if (rc_eval) return -1;
    return gs_grad_post(ctx, sl, sl->ws, desc, params, P, solo, solo && ctx->last_potrf_chain,
                        ctx->grad_run.prepared ? &ctx->grad_run : nullptr);
}

// everything behind the factorisation of m (L in m->A, the tables of its diagonal blocks, W^T in the border rows), on the slot's streams
static int gs_grad_post(gsum_ctx* ctx, gs_slot* sl, gsum_mat* m, const gsum_kernel_desc* desc, const gsum_grad_param* params, int P, bool solo,
                        bool on_chain, gs_grad_run* started) {
    ctx->cur = sl;
    gs_grad_run local;
    gs_grad_run* r = started;
    if (!r) {
        r = &local;
        if (gs_grad_reserve(ctx, sl, ctx->in->n, m->np, P, solo)) return -1;
        if (gs_grad_prepare(ctx, r, sl, m, P, solo, on_chain)) return -1;
    } else if (r->m != m || !on_chain) {
        GS_FAIL("internal: the sweep was started for another factorisation");
    }
    if (gs_grad_sweep(ctx, r, m->T)) return -1;
    const int64_t n = r->n, np = r->np, ld = r->ld, ldg = r->ldg;
    const int d = r->d, chunks = r->chunks, rows_per = r->rows_per;
    double *U = r->U, *Ri = r->Ri, *Vt = r->Vt, *Q = r->Q, *trow = r->trow, *dout = r->dout, *part = r->part;
    hipStream_t s = r->s, su = r->su;
    hipStream_t sv = s;
    if (solo) {
        GS_CHECK(hipEventRecord(sl->evU, su));
        GS_CHECK(hipStreamWaitEvent(s, sl->evU, 0));
        // V^T = W^T U^T needs only U: it runs on the panel stream beside the SYRK
        GS_CHECK(hipEventRecord(sl->evFork, s));
        if (gs_panel_stream(ctx, sl)) return -1;
        GS_CHECK(hipStreamWaitEvent(sl->sp, sl->evFork, 0));
        sv = sl->sp;
    }
    hipLaunchKernelGGL(k_upper_times_rows, dim3((unsigned)((np + 15) / 16)), dim3(256), 0, sv, U, ldg, (int)np, m->A + np * ld, ld, Vt, ldg);
    GS_CHECK(hipGetLastError());
    gs_grad_params prm;
    memset(&prm, 0, sizeof prm);
    for (int p = 0; p < P; ++p) prm.p[p] = params[p];
    // One evaluation alone (round 5): Q_p = dR_p V -- every kernel-gradient evaluation of the stage, vector-pipe work that needs V^T only -- runs
    // behind V^T on the panel stream BESIDE the product R^-1 = U U^T (matrix pipes), leaves the lower triangles of dR_p in memory, and the traces
    // are a streaming pass over R^-1 and those triangles afterwards: 1.1 ms of contractions behind the product become ~0.1.
    const bool split = r->split;
    if (split) {
        if (desc->n_ops > 0) {
            hipLaunchKernelGGL((k_grad_contract<true, 1, true>), dim3((unsigned)((n + 3) / 4), (unsigned)P), dim3(256), 0, sv, ctx->in->X, (int)n, (int)d,
                               *desc, prm, Ri, ldg, Vt, ldg, Q, trow, r->dR, r->dr_stride);
        } else {
            hipLaunchKernelGGL((k_grad_contract<false, 2, true>), dim3((unsigned)((n + 7) / 8), (unsigned)P), dim3(256), 0, sv, ctx->in->X, (int)n, (int)d,
                               *desc, prm, Ri, ldg, Vt, ldg, Q, trow, r->dR, r->dr_stride);
        }
        GS_CHECK(hipGetLastError());
    }
    if (solo) {
        if (gs_potrf_events(ctx, sl, 1)) return -1;
        GS_CHECK(hipEventRecord(sl->evP[0], sl->sp));
    }
    if (gs_gemm(ctx, s, GS_BULK, Ri, ldg, U, ldg, U, ldg, np, np, (int)np, 2, 0, 1.0)) return -1;
    if (solo) GS_CHECK(hipStreamWaitEvent(s, sl->evP[0], 0));
    if (split) {
        hipLaunchKernelGGL(k_grad_trace, dim3((unsigned)((n + 3) / 4), (unsigned)P), dim3(256), 0, s, Ri, ldg, r->dR, r->dr_stride, (int)n, trow);
    } else if (desc->n_ops > 0) {
        hipLaunchKernelGGL((k_grad_contract<true, 1>), dim3((unsigned)((n + 3) / 4), (unsigned)P), dim3(256), 0, s, ctx->in->X, (int)n, (int)d, *desc,
                       prm, Ri, ldg, Vt, ldg, Q, trow);
    } else {
        hipLaunchKernelGGL((k_grad_contract<false, 2>), dim3((unsigned)((n + 7) / 8), (unsigned)P), dim3(256), 0, s, ctx->in->X, (int)n, (int)d, *desc,
                       prm, Ri, ldg, Vt, ldg, Q, trow);
    }
    GS_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_grad_reduce1, dim3((unsigned)chunks, (unsigned)P), dim3(256), 0, s, Vt, ldg, Q, trow, (int)n, rows_per, part);
    GS_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_grad_reduce2, dim3((unsigned)P), dim3(256), 0, s, part, chunks, dout);
    GS_CHECK(hipGetLastError());
    GS_CHECK(hipMemcpyAsync(sl->hgrad, dout, (size_t)P * 257 * 8, hipMemcpyDeviceToHost, s));
    return 0;
}

// wait for the gradient evaluation pending on a slot (index sl->pending) and copy its pieces out
static int gs_grad_harvest(gsum_ctx* ctx, gs_slot* sl, int P, double* G_out, double* sld_out, int64_t* info_out, double* trace_out,
                           double* H_out) {
    const int i = sl->pending, k = ctx->in->k;
    if (i < 0) return 0;
    if (gs_eval_harvest(ctx, sl, G_out, sld_out, info_out)) return -1;      // synchronises the stream
    for (int p = 0; p < P; ++p) {
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) H_out[(((size_t)i * P + p) * k + a) * k + b] = sl->hgrad[(size_t)p * 257 + a * 16 + b];
        trace_out[(size_t)i * P + p] = sl->hgrad[(size_t)p * 257 + 256];
    }
    return 0;
}

// n <= GS_GSMALL_MAX: value + gradient pieces of every kernel of the call in ONE launch, one workgroup each (k_grad_small): the reference's own
// sizes, where a dozen launches and two synchronisations per objective evaluation cost more than the arithmetic.  Measured per objective
// evaluation through the class (tools/gpu_small_fit_profile.py): n = 8 / 20 / 48 / 64 / 96 / 128: 189 / 194 / 211 / 231 / 276 / 333 us against
// ~325-340 on the general path (the first form of the kernel walked the contractions' rows with one wave per two rows and lost above n = 64;
// since the contractions are matrix products the one-block orders are all its).
#define GS_GSMALL_MAX 128
static int gs_grad_small(gsum_ctx* ctx, const gsum_kernel_desc* descs, int n_desc, const gsum_grad_param* params, int P, double nugget,
                         double* G_out, double* sld_out, int64_t* info_out, double* trace_out, double* H_out) {
    const int k = ctx->in->k, CH = std::min(512, n_desc);
    ctx->cur = &ctx->slots[0];
    hipStream_t s = ctx->cur->sm;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    // descriptors and gradient parameters travel in ONE copy, results and gradient pieces come back in ONE (each copy of a pageable buffer is
    // ~10 us of a ~100-us call): [descs | params] and [res | gres] are contiguous on the device, offsets by the launch's count
    const size_t in_cap = (size_t)CH * (sizeof(gsum_kernel_desc) + (size_t)P * sizeof(gsum_grad_param));
    const size_t out_cap = (size_t)CH * (258 + (size_t)P * 257) * 8;
    const size_t o_in = 0, o_out = up(in_cap), o_scr = o_out + up(out_cap);
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, o_scr + (size_t)CH * GS_GSMALL_SCRATCH * 8)) return -1;
    char* base = (char*)ctx->scratch;
    if (gs_reserve_pinned(ctx, up(in_cap) + out_cap)) return -1;            // (staged inputs | results: two regions, no wait in between)
    for (int lo = 0; lo < n_desc; lo += CH) {
        const int cnt = std::min(CH, n_desc - lo);
        const size_t dbytes = (size_t)cnt * sizeof(gsum_kernel_desc), pbytes = (size_t)cnt * P * sizeof(gsum_grad_param);
        char* stage = (char*)ctx->hbatch;                      // (pinned; the previous launch's copy out of it completed before its results were unpacked)
        memcpy(stage, descs + lo, dbytes);
        memcpy(stage + dbytes, params + (size_t)lo * P, pbytes);
        GS_CHECK(hipMemcpyAsync(base + o_in, stage, dbytes + pbytes, hipMemcpyHostToDevice, s));
        bool tree = false;
        for (int e = 0; e < cnt; ++e) tree = tree || descs[lo + e].n_ops > 0;
        double* dres = (double*)(base + o_out);
        double* dgres = dres + (size_t)cnt * 258;
        hipLaunchKernelGGL(tree ? k_grad_small<true> : k_grad_small<false>, dim3(cnt), dim3(256), 0, s, ctx->in->X, (int)ctx->in->n, ctx->in->d,
                           ctx->in->Z, k, (const gsum_kernel_desc*)(base + o_in), (const gsum_grad_param*)(base + o_in + dbytes), P, nugget,
                           (double*)(base + o_scr), dres, dgres);
        GS_CHECK(hipGetLastError());
        double* hres = (double*)((char*)ctx->hbatch + up(in_cap));
        GS_CHECK(hipMemcpyAsync(hres, dres, (size_t)cnt * (258 + (size_t)P * 257) * 8, hipMemcpyDeviceToHost, s));
        GS_CHECK(hipStreamSynchronize(s));
        const double* hg = hres + (size_t)cnt * 258;
        for (int e = 0; e < cnt; ++e) {
            const double* r = hres + (size_t)e * 258;
            const int i = lo + e;
            for (int a = 0; a < k; ++a)
                for (int b = 0; b < k; ++b) G_out[(size_t)i * k * k + a * k + b] = r[a * 16 + b];
            sld_out[i] = r[256];
            info_out[i] = (int64_t)r[257];
            for (int p = 0; p < P; ++p) {
                const double* g = hg + ((size_t)e * P + p) * 257;
                const bool ok = info_out[i] == 0;              // not positive definite: no gradient pieces (the caller looks at info)
                for (int a = 0; a < k; ++a)
                    for (int b = 0; b < k; ++b) H_out[(((size_t)i * P + p) * k + a) * k + b] = ok ? g[a * 16 + b] : 0.0;
                trace_out[(size_t)i * P + p] = ok ? g[256] : 0.0;
            }
        }
    }
    return 0;
}

// one value + gradient evaluation alone on slot 0 (inputs already uploaded, descriptor checked)
static int gs_grad_single(gsum_ctx* ctx, const gsum_kernel_desc* desc, const gsum_grad_param* params, int n_params, double nugget,
                          double* G_out, double* sld_out, int64_t* info_out, double* trace_out, double* H_out) {
    gs_slot* sl = &ctx->slots[0];
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int aborts = ctx->chain_aborts;
        ctx->batch_active = 1;
        if (gs_grad_enqueue(ctx, sl, desc, params, n_params, nugget, true)) return -1;
        sl->pending = 0;
        const int rc = gs_grad_harvest(ctx, sl, n_params, G_out, sld_out, info_out, trace_out, H_out);
        // a persistent chain that gave up: gs_eval_harvest has re-run the VALUE on the host-enqueued schedule (and switched the chain
        // schedule off), but the gradient pieces came from the abandoned factorisation -- once more, now without the chain
        if (rc || ctx->chain_aborts == aborts) return rc;
    }
    return 0;
}

int gsum_lml_grad(gsum_ctx* ctx, const gsum_kernel_desc* desc, const gsum_grad_param* params, int32_t n_params,
                  const double* X, int64_t n, int32_t d, const double* RHS, int32_t k, double nugget, double* G_out,
                  double* sld_out, int64_t* info_out, double* trace_out, double* H_out) {
    if (!ctx || !desc || !params || !G_out || !sld_out || !info_out || !trace_out || !H_out) return -2;
    if (gs_grad_check(ctx, params, n_params, d, k)) return -2;
    if (gs_check_order(ctx, n)) return -2;
    int rc = gs_upload_inputs(ctx, X, n, d, RHS, k);
    if (rc) return rc;
    ctx->in = &ctx->op;
    if (gs_check_desc(ctx, desc, d)) return -2;
    if (n <= GS_GSMALL_MAX && ctx->small_path) return gs_grad_small(ctx, desc, 1, params, n_params, nugget, G_out, sld_out, info_out, trace_out, H_out);
    return gs_grad_single(ctx, desc, params, n_params, nugget, G_out, sld_out, info_out, trace_out, H_out);
}

// A batch of value + gradient evaluations on the GROUPED schedule (round 4): the factorisations of up to wave_groups x wave_size kernels
// run as one gs_lml_wave call (3.0 ms each at n = 8192 instead of ~5 on a stream of their own), which leaves every member's factor, tables
// and solved border rows in its workspace; the gradient stage of each member (U = L^-T, R^-1 = U U^T, V^T, the contractions: 2 n^3 / 3 flops)
// then runs on one of the context's four streams, four members at a time.  Same kernels on the same data as the one-stream-per-evaluation
// path: bit-identical results.
static int gs_grad_batch_wave(gsum_ctx* ctx, const gsum_kernel_desc* descs, int n_desc, const gsum_grad_param* params, int P, double nugget,
                              double* G_out, double* sld_out, int64_t* info_out, double* trace_out, double* H_out) {
    const int k = ctx->in->k;
    const int64_t n = ctx->in->n;
    const int S = std::max(1, std::min(4, ctx->batch_slots));
    if (gs_need_slots(ctx, S)) return -1;
    // the slots' gradient buffers first (U, R^-1, ...: 3.2 n^2 doubles each), then as many members per chunk as ONE round of the groups holds
    {
        // (the wave pool pads to 256, gs_padded_order to 128 when the chain schedule is off or the order small: the larger of the two, or
        // gs_grad_post would re-reserve -- hipFree + hipMalloc, a device-wide synchronisation -- per slot inside the batch)
        const int64_t npg = std::max<int64_t>(gs_padded_order(ctx, n), (n + 2 * GS_NB - 1) / (2 * GS_NB) * (2 * GS_NB));
        const size_t want = (size_t)(3.3 * (double)(npg + GS_BORDER) * (double)(npg + GS_BORDER) * 8.0);
        for (int q = 0; q < S; ++q)
            if (gs_reserve(ctx, &ctx->slots[q].gws, &ctx->slots[q].gws_cap, want)) return -1;
    }
    const int64_t npw = (n + 2 * GS_NB - 1) / (2 * GS_NB) * (2 * GS_NB);
    const int fit = gs_wave_fit(ctx, n, npw);
    if (fit < 1) GS_FAIL("not enough device memory for one workspace matrix");
    const int Gq = std::max(1, std::min(std::min(GS_WV_STREAM_GROUPS, ctx->wave_groups), fit));
    const int cap = Gq * std::max(1, std::min(std::min(GS_WVC_MAX, ctx->wave_size), fit / Gq));       // a multiple of the groups: ONE round per chunk
    auto harvest = [&](gs_slot* sl) -> int {                 // wait for the gradient stage pending on a slot, copy its pieces out
        const int i = sl->pending;
        if (i < 0) return 0;
        GS_CHECK(hipStreamSynchronize(sl->sm));
        for (int p = 0; p < P; ++p) {
            for (int a = 0; a < k; ++a)
                for (int b = 0; b < k; ++b) H_out[(((size_t)i * P + p) * k + a) * k + b] = sl->hgrad[(size_t)p * 257 + a * 16 + b];
            trace_out[(size_t)i * P + p] = sl->hgrad[(size_t)p * 257 + 256];
        }
        sl->pending = -1;
        return 0;
    };
    for (int off = 0; off < n_desc; off += cap) {
        const int cnt = std::min(cap, n_desc - off);
        ctx->cur = &ctx->slots[0];
        if (gs_lml_wave(ctx, descs + off, cnt, nugget, G_out + (size_t)off * k * k, sld_out + off, info_out + off)) return -1;
        // one round: group g holds members [first_eval, first_eval + cnt) of this chunk, member q at workspace q of its pool
        {
            int covered = 0;
            for (int gi = 0; gi < GS_WV_GROUPS; ++gi)
                if (ctx->wave.g[gi].n == n) covered += ctx->wave.g[gi].cnt;
            if (covered != cnt) GS_FAIL("internal: a chunk of the gradient batch took more than one round of the groups");
        }
        ctx->batch_active = std::max(S, 3);
        int rc = 0, slot = 0;
        for (int gi = 0; gi < GS_WV_GROUPS && !rc; ++gi) {
            gs_wave_group* g = &ctx->wave.g[gi];
            if (g->cnt <= 0 || g->n != n) continue;
            for (int q = 0; q < g->cnt && !rc; ++q) {
                const int i = off + g->first_eval + q;
                if (i >= off + cnt) break;
                if (info_out[i] != 0) {                       // not positive definite: no gradient pieces (the caller looks at info)
                    for (int p = 0; p < P; ++p) {
                        trace_out[(size_t)i * P + p] = 0.0;
                        for (int ab = 0; ab < k * k; ++ab) H_out[((size_t)i * P + p) * k * k + ab] = 0.0;
                    }
                    continue;
                }
                gs_slot* sl = &ctx->slots[slot++ % S];
                rc = harvest(sl);
                if (rc) break;
                gsum_mat view;
                view.n = n;
                view.np = g->pool.np;
                view.ld = g->pool.ld;
                view.T = g->pool.T;
                view.A = g->pool.A + (int64_t)q * g->pool.strideA;
                view.Ltab = g->pool.Ltab + (size_t)q * g->pool.T * GS_LTAB;
                view.have_ltab = true;
                view.Lsib = g->pool.Lsib + (size_t)q * (g->pool.T / 2 + 1) * GS_LSIB;      // (left by k_potrf_diag256g: the grouped factorisation)
                view.have_lsib = true;
                rc = gs_grad_post(ctx, sl, &view, &descs[i], params + (size_t)i * P, P, false, false, nullptr);
                if (!rc) sl->pending = i;
            }
        }
        for (int q = 0; q < S; ++q) {
            const int r2 = harvest(&ctx->slots[q]);
            if (!rc) rc = r2;
        }
        // (the groups' counts describe the last call only: clear them so that a later reader cannot take them for current)
        ctx->cur = &ctx->slots[0];
        ctx->batch_active = 1;
        if (rc) return rc;
    }
    return 0;
}

// The same for a list of kernels with ONE hyperparameter structure (params: n_desc x n_params entries, the weights are per kernel) on one set of inputs (the restarts of a multi-start fit,
// models.py:641-662; a grid of gradients): independent evaluations pipelined over slots like gsum_lml_resident's, each entirely on
// its slot's main stream.  Outputs are the single-evaluation outputs stacked: G (n, k, k), sld (n), info (n), trace (n, P), H (n, P, k, k).
int gsum_lml_grad_batch(gsum_ctx* ctx, const gsum_kernel_desc* descs, int32_t n_desc, const gsum_grad_param* params, int32_t n_params,
                        const double* X, int64_t n, int32_t d, const double* RHS, int32_t k, double nugget, double* G_out,
                        double* sld_out, int64_t* info_out, double* trace_out, double* H_out) {
    if (!ctx || !descs || !params || !G_out || !sld_out || !info_out || !trace_out || !H_out || n_desc < 1) return -2;
    for (int i = 0; i < n_desc; ++i) {
        if (gs_grad_check(ctx, params + (size_t)i * n_params, n_params, d, k)) return -2;
        for (int p = 0; p < n_params; ++p)
            if (params[(size_t)i * n_params + p].code != params[p].code || params[(size_t)i * n_params + p].dim != params[p].dim)
                GS_FAIL("gsum_lml_grad_batch: every kernel must have the same hyperparameter structure");
    }
    if (gs_check_order(ctx, n)) return -2;
    int rc = gs_upload_inputs(ctx, X, n, d, RHS, k);
    if (rc) return rc;
    ctx->in = &ctx->op;
    for (int i = 0; i < n_desc; ++i)
        if (gs_check_desc(ctx, &descs[i], d)) return -2;
    if (n <= GS_GSMALL_MAX && ctx->small_path) return gs_grad_small(ctx, descs, n_desc, params, n_params, nugget, G_out, sld_out, info_out, trace_out, H_out);
    if (n_desc == 1) return gs_grad_single(ctx, &descs[0], params, n_params, nugget, G_out, sld_out, info_out, trace_out, H_out);
    if (n_desc >= ctx->wave_min && n > 256 && ctx->grad_batch_wave)
        return gs_grad_batch_wave(ctx, descs, n_desc, params, n_params, nugget, G_out, sld_out, info_out, trace_out, H_out);
    // slots: every one owns a workspace matrix and U, R^-1 (3 n^2 doubles in all): within 70 % of the free memory, 8 at most
    const int64_t np = gs_padded_order(ctx, n);
    size_t free_b = 0, total_b = 0;
    GS_CHECK(hipMemGetInfo(&free_b, &total_b));
    const double per_slot = 3.2 * (double)(np + GS_BORDER) * (double)(np + GS_BORDER) * 8.0;
    int S = (int)std::min<double>(8.0, std::max(1.0, 0.7 * (double)free_b / per_slot));
    S = std::max(1, std::min(S, (int)n_desc));
    S = std::min(S, ctx->batch_slots);                              // one stream each: the context's four streams sit on four pipes
    if (gs_need_slots(ctx, S)) return -1;
    ctx->batch_active = std::max(S, 3);        // the batch schedule (no look-ahead, no intra-evaluation events) for every member
    for (int i = 0; i < n_desc && !rc; ++i) {
        gs_slot* sl = &ctx->slots[i % S];
        rc = gs_grad_harvest(ctx, sl, n_params, G_out, sld_out, info_out, trace_out, H_out);
        if (!rc) rc = gs_grad_enqueue(ctx, sl, &descs[i], params + (size_t)i * n_params, n_params, nugget, false);
        if (!rc) sl->pending = i;
    }
    for (int q = 0; q < S; ++q) {
        const int r2 = gs_grad_harvest(ctx, &ctx->slots[q], n_params, G_out, sld_out, info_out, trace_out, H_out);
        if (!rc) rc = r2;
    }
    ctx->cur = &ctx->slots[0];
    ctx->batch_active = 1;
    return rc;
}

