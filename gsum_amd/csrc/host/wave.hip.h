// the grouped batch schedule (gs_lml_wave)
// (part of gsum_capi.hip: included from there, in order -- one translation unit)
#pragma once
// ---- grouped batch schedule ---------------------------------------------------------------------------------------------------------
// The evaluations of one call are independent (a likelihood grid, gsum/models.py:958-1039 per grid point; the reference's loop is
// docs/notebooks/correlated_EFT_publication.ipynb:1457-1459).  They are cut into groups of up to `wave_size`; the members of a group
// advance through the outer steps of the blocked factorisation in lock step, and one launch per kernel class carries that step for
// all of them (k_potrf_diag256g: a workgroup per member; k_panel256g: a wave per 16 rows of every member; k_gemm_ld3g: the tiles of
// every member's trailing update).  Streams: one chain stream per group + ONE bulk stream; per group and step
//     chain stream:  [bulk(g, s - 1) done]  diag(g, s)  panel(g, s)          -> evChain
//     bulk stream:   [evChain]  bulk(g, s)                                     -> evBulk
// and the host enqueues the groups round-robin, so that on the bulk stream the trailing updates of the groups alternate while the
// latency-bound chain of one group runs beside the trailing update of the other(s).  Nothing depends on how many hardware queues the
// runtime was started with (3 streams for two groups), a rank under torch.distributed.run runs the same schedule as a lone process,
// and a per-launch profile IS the step time: the bulk launches do not overlap one another.
// Per element of every matrix the same products are subtracted in the same order as in the one-stream-per-evaluation schedule
// (same kernels' bodies, same K = 256 / K = 512 pairing of the trailing updates): G, sum log L_ii and info are bit-identical to it.
static void gs_wave_free_group(gs_wave_group* g) {
    for (void* q : {(void*)g->pool.A, (void*)g->pool.Ltab, (void*)g->pool.Lsib, (void*)g->pool.logdet, (void*)g->pool.diag0,
                    (void*)g->pool.info, (void*)g->pool.res})
        if (q) (void)hipFree(q);
    memset(&g->pool, 0, sizeof g->pool);
    g->cap = 0;
    g->n = 0;
}

static void gs_wave_release(gsum_ctx* ctx, bool streams) {
    for (int i = 0; i < GS_WV_GROUPS; ++i) {
        gs_wave_group* g = &ctx->wave.g[i];
        gs_wave_free_group(g);
        if (!streams) continue;
        if (g->sc && g->own_sc) (void)hipStreamDestroy(g->sc);
        g->own_sc = false;
        if (g->evChain) (void)hipEventDestroy(g->evChain);
        if (g->evBulk) (void)hipEventDestroy(g->evBulk);
        g->sc = nullptr;
        g->evChain = g->evBulk = nullptr;
    }
    if (streams) ctx->wave.sb = nullptr;             // (slot 0's main stream: not the groups' to destroy)
}

static double gs_wave_ws_bytes(int64_t np) {
    const double T = (double)(np / GS_NB);
    return (double)(np + GS_BORDER) * (double)GS_LD(np) * 8.0 + T * GS_LTAB * 8.0 + (T / 2 + 1) * GS_LSIB * 8.0 +
           (T + (double)np + 258.0) * 8.0 + 4.0;
}

// G groups in all, the first Gs of them with a chain stream of their own; group i >= Gs (a second cohort) runs on group (i - Gs)'s
static int gs_wave_prepare(gsum_ctx* ctx, int G, int B, int64_t n, int64_t np, int Gs) {
    gs_wave* wv = &ctx->wave;
    if (!wv->sb) wv->sb = ctx->slots[0].sm;
    const int T = (int)(np / GS_NB);
    const int64_t ld = GS_LD(np);
    for (int i = 0; i < G; ++i) {
        gs_wave_group* g = &wv->g[i];
        if (i >= Gs) g->run = wv->g[i - Gs].sc;            // a second cohort: on the stream of the group it shadows (ensured above: i - Gs < i)
        if (i < Gs && !g->sc) {
            // A batch call and a single factorisation never run at the same time: the first two groups run on slot 0's two
            // high-priority streams, the third on the stream gsum_init created next to them (see there: four streams on four pipes)
            gs_slot* s0 = &ctx->slots[0];
            if (i == 0) { if (gs_panel_stream(ctx, s0)) return -1; g->sc = s0->sp; }
            else if (i == 1) { if (gs_aux_stream(ctx, s0)) return -1; g->sc = s0->sa; }
            else {
                GS_CHECK(hipStreamCreateWithPriority(&g->sc, hipStreamNonBlocking, ctx->prio_hi));
                g->own_sc = true;
                // a stream created now (a fourth group) sits on a pipe one of the first four already uses: probe again, so that
                // "pipes_ok" describes the streams this batch runs on
                hipStream_t all[GS_WV_GROUPS + 1] = {wv->sb};
                int cnt = 1;
                for (int q = 0; q <= i; ++q) all[cnt++] = wv->g[q].sc;
                if (gs_pipe_probe(ctx, all, cnt, nullptr)) return -1;
            }
        }
        if (i < Gs) g->run = g->sc;
        if (!g->evChain) {
            GS_CHECK(hipEventCreateWithFlags(&g->evChain, hipEventDisableTiming));
            GS_CHECK(hipEventCreateWithFlags(&g->evBulk, hipEventDisableTiming));
        }
        if (g->cap >= B && g->n == n) continue;
        GS_CHECK(hipDeviceSynchronize());
        gs_wave_free_group(g);
        gs_wv_pool& p = g->pool;
        p.strideA = (np + GS_BORDER) * ld;
        p.ld = ld;
        p.np = (int)np;
        p.T = T;
        hipError_t e = hipMalloc((void**)&p.A, (size_t)B * p.strideA * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void**)&p.Ltab, (size_t)B * T * GS_LTAB * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void**)&p.Lsib, (size_t)B * (T / 2 + 1) * GS_LSIB * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void**)&p.logdet, (size_t)B * T * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void**)&p.diag0, (size_t)B * np * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void**)&p.info, (size_t)B * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void**)&p.res, (size_t)B * 258 * sizeof(double));
        if (e != hipSuccess) {
            gs_wave_free_group(g);
            ctx->err = std::string("hipMalloc(group workspaces) failed: ") + hipGetErrorString(e);
            return -1;
        }
        g->cap = B;
        g->n = n;
    }
    return 0;
}

// Trailing updates of a batch, per outer step s (panel columns [256 s, 256 s + 256), trailing matrix from r2 = 256 (s + 1)).  Steps
// are grouped into macro-steps of up to `depth` panels a .. a + L - 1 (L panels are grouped only while r2(a) + 256 L <= np):
//   step a + i, i < L - 1   "near": only the NEXT panel's 256 columns are updated, with all panels of the macro-step so far at once
//                           (rows r2.., rectangular, K = 256 (i + 1)) -- what the chain's next link needs;
//   step a + L - 1          "far": everything from column r2 on, lower tiles, with all L panels in ONE pass (K = 256 L).
// depth 1: a plain right-looking sweep (K = 256 everywhere); depth 2: the pairing of gs_potrf's batch branch (lazy_far = 2).  A deeper
// grouping reads and writes the far region once per L panels -- the bulk tile's rate rises with K (C traffic per flop) -- at the
// price of near updates with K up to 256 (L - 1).  Per element the same products are subtracted in the same ascending order whatever
// the grouping (an accumulator that starts as C carries across launches exactly): results do not depend on it.
struct gs_wave_step { int near; int K; int first; };        // first: the macro-step's first outer step (the operand's first panel)
// first_len > 0: the FIRST macro-step has at most that many panels (the head of a call: see gs_lml_wave).
static void gs_wave_bulk_plan(int64_t np, int depth, int64_t deep_min_rows, int first_len, std::vector<gs_wave_step>& plan) {
    const int S = (int)(np / (2 * GS_NB));
    plan.assign((size_t)S, gs_wave_step{0, 2 * GS_NB, 0});
    for (int a = 0; a < S;) {
        const int64_t r2 = 2 * GS_NB * (int64_t)(a + 1);
        int L = 1;
        while (L < depth && r2 + 2 * GS_NB * (int64_t)(L + 1) <= np) ++L;
        if (L > 2 && np + GS_BORDER - r2 < deep_min_rows) L = 2;          // deeper than pairs only while the trailing matrix is large
        if (a == 0 && first_len > 0) L = std::min(L, first_len);
        for (int i = 0; i < L; ++i) plan[(size_t)(a + i)] = gs_wave_step{i < L - 1 ? 1 : 0, 2 * GS_NB * (i + 1), a};
        a += L;
    }
}

static int gs_wave_fill_chain(const gs_wave_group* g, gs_wv_chain_args* a, bool panel_counts) {
    a->p = g->pool;
    a->n = g->cnt;
    a->pad = 0;
    const int naug = g->pool.np + GS_BORDER;
    int run = 0;
    for (int e = 0; e < g->cnt; ++e) {
        a->q[e] = (short)e;
        a->step[e] = (short)g->step;
        if (panel_counts) run += (naug - 2 * GS_NB * (g->step + 1)) / 16;
        a->end[e] = run;
    }
    return run;
}

// how many workspace matrices of this order the groups may hold: within 70 % of what is free (plus what the groups already hold)
static int gs_wave_fit(gsum_ctx* ctx, int64_t n, int64_t np) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
    double held = 0.0;
    for (int i = 0; i < GS_WV_GROUPS; ++i)
        if (ctx->wave.g[i].cap && ctx->wave.g[i].n == n) held += ctx->wave.g[i].cap * gs_wave_ws_bytes(np);
    return (int)std::min<double>(1e6, (0.7 * (double)free_b + held) / gs_wave_ws_bytes(np));
}

static int gs_lml_wave(gsum_ctx* ctx, const gsum_kernel_desc* kernels, int n_kernels, double nugget, double* G_out, double* sld_out,
                       int64_t* info_out) {
    const int64_t n = ctx->in->n, np = (n + 2 * GS_NB - 1) / (2 * GS_NB) * (2 * GS_NB), ld = GS_LD(np), naug = np + GS_BORDER;
    const int k = ctx->in->k, d = ctx->in->d, S = (int)(np / (2 * GS_NB));
    int G = std::max(1, std::min(GS_WV_STREAM_GROUPS, ctx->wave_groups));
    int B = std::max(1, std::min(GS_WVC_MAX, ctx->wave_size));
    if (n_kernels < G * B) {                      // a short call: every evaluation in flight at once, the groups equally full
        G = std::min(G, n_kernels);
        B = (n_kernels + G - 1) / G;
    }
    const int Gs = G;                             // groups with a stream of their own
    // TWO COHORTS per group in calls of many rounds (round 5): rounds in phase end with ~2.5 ms of latency-bound last steps in which
    // no group has a far update worth the chip (4 % of a 71-ms round at n = 8192), and groups out of phase on streams of their own lose
    // more than that (a group's chain is then covered by ONE other group's update instead of two: measured, tools/gpu_long_call.py).  So
    // every group gets a second set of workspaces and runs a second cohort of evaluations half a round behind the first ON THE SAME
    // chain stream: per sweep of the loop below the stream carries cohort one's macro-step, then cohort two's, and the bulk stream
    // both far updates -- the last steps of one cohort run under the far updates of the other, with no stream added.
    const bool cohorts = ctx->wave_cohorts >= 2 && n_kernels >= ctx->wave_cohort_min * Gs * B && 2 * Gs <= GS_WV_GROUPS;
    if (cohorts) G = 2 * Gs;
    {
        const int fit = gs_wave_fit(ctx, n, np);
        if (fit < 1) GS_FAIL("not enough device memory for one workspace matrix");
        if (G * B > fit) {
            if (cohorts && Gs * B <= fit) B = std::max(1, fit / G);          // (both cohorts, smaller)
            else {
                G = std::max(1, std::min(G, fit));
                B = std::max(1, fit / G);
            }
        }
    }
    // A call of several rounds hands out EQUAL shares: R = ceil(n / (G B)) rounds, G R group-rounds of floor or ceil(n / (G R))
    // evaluations each (64 evaluations on 3 x 8: nine group-rounds of 7 or 8 -- not nine of 7 and a tenth with ONE evaluation
    // running alone at the end, which is what first-come-first-served refills did to the 64-per-call scan of bench.py).
    std::vector<int> shares;
    {
        const int R = (n_kernels + G * B - 1) / (G * B), parts = G * R;
        for (int p = 0; p < parts; ++p) shares.push_back(n_kernels / parts + (p < n_kernels % parts ? 1 : 0));
        B = std::min(B, shares[0]);
    }
    size_t next_share = 0;
    if (gs_wave_prepare(ctx, G, B, n, np, std::min(Gs, G))) return -1;
    ctx->wave_last_streams = std::min(Gs, G) + 1;
    if (gs_reserve_pinned(ctx, (size_t)n_kernels * 258 * sizeof(double))) return -1;
    gs_wave* wv = &ctx->wave;
    // One plan per group: in a call's first round the groups' first macro-steps differ in length (option wave_head, decimal digits,
    // one per group) -- with every group four panels deep, the bulk stream's first update starts only after four chain steps (1.6 ms
    // of a 61-ms call at n = 8192).  Results do not depend on the grouping.
    std::vector<gs_wave_step> plans[GS_WV_GROUPS];
    {
        // (grouping from padded order 1024 up: round 3's threshold of 4352 belonged to the one-stream-per-evaluation batch; on the grouped
        //  schedule 24 evaluations at n = 4096 take 12.5 ms with it and 13.7 without, n = 3072: 6.1 / 6.6, n = 2048: 3.1 / 3.2, n = 1024: the same)
        const int depth = (ctx->lazy_far != 0 && np >= std::min(ctx->lazy_min_np, 1024)) ? std::max(2, ctx->wave_depth) : 1;
        int digits[GS_WV_GROUPS] = {0};
        int h = ctx->wave_head, nd = 0;
        int tmp[8];
        while (h > 0 && nd < 8) { tmp[nd++] = h % 10; h /= 10; }
        for (int i = 0; i < GS_WV_GROUPS; ++i) digits[i] = i < nd ? tmp[nd - 1 - i] : 0;
        for (int i = 0; i < GS_WV_GROUPS; ++i) gs_wave_bulk_plan(np, depth, ctx->wave_deep_rows, depth > 1 ? digits[i] : 0, plans[i]);
    }
    const bool several_rounds = n_kernels > G * B;
    // Groups out of phase in calls of several rounds (counted in sweeps of the loop below = macro-steps)?  Measured and left off:
    // the groups' big updates alternate on one stream, so all groups advance at the same macro-step rate, and a group in its
    // latency-bound last steps is paced by the other's 5-ms updates; 80 evaluations on 2 x 10: 315 evals/s in phase, 305 / 300
    // with the second group 4 / 8 macro-steps behind (the chains of the first and last macro-steps then run with nothing beside them).
    const int rounds_of_call = (n_kernels + G * B - 1) / (G * B);
    const bool long_call = rounds_of_call >= ctx->wave_long_rounds;
    // calls of many rounds (a whole likelihood surface in one call, gsum_lml_resident_sets): the groups run out of phase by a third (1 / G)
    // of a round's macro-steps and the far updates of a group's last macro-steps stay on its own stream (wave_shift = -1: automatic)
    int ticks_per_round = 0;
    for (int st_ = 0; st_ < S; ++st_) ticks_per_round += plans[0][(size_t)st_].near ? 0 : 1;
    const int auto_shift = long_call ? std::max(1, ticks_per_round / G) : 0;
    const int shift = !several_rounds ? 0 : (ctx->wave_shift > 0 ? std::min(ctx->wave_shift, S) : (ctx->wave_shift < 0 ? auto_shift : 0));
    const int64_t tail_rows = (several_rounds && (cohorts || ctx->wave_tail_rows < 0)) ? std::abs(ctx->wave_tail_rows) : 0;
    for (int i = 0; i < GS_WV_GROUPS; ++i) {           // (all groups: after the call cnt / first_eval say which members a group's workspaces hold)
        gs_wave_group* g = &wv->g[i];
        g->active = false;
        g->cnt = g->step = 0;
        g->start_tick = i * shift;
        if (cohorts && i >= Gs && i < G) g->start_tick = std::max(1, ticks_per_round / 2);      // the second cohorts: half a round behind
    }
    const bool prof = ctx->profile_gemm > 0;
    if (prof) ctx->prof_this_eval = true;
    // everything of this call follows what the context's main stream has done so far (the resident inputs' upload)
    hipStream_t s0 = ctx->slots[0].sm;
    GS_CHECK(hipEventRecord(ctx->slots[0].evFork, s0));
    GS_CHECK(hipStreamWaitEvent(wv->sb, ctx->slots[0].evFork, 0));
    for (int i = 0; i < G; ++i) GS_CHECK(hipStreamWaitEvent(wv->g[i].run, ctx->slots[0].evFork, 0));
    int next = 0, live = 0;
    for (int tick = 0; next < n_kernels || live > 0; ++tick) {
        for (int i = 0; i < G; ++i) {
            gs_wave_group* g = &wv->g[i];
            gs_wv_chain_args ca;
            if (!g->active) {
                if (next >= n_kernels || tick < g->start_tick) continue;
                // ---- a new round of this group: its next evaluations enter (their workspaces are free: the read-out of the
                // previous round is ahead of this on the chain stream)
                g->cnt = next_share < shares.size() ? shares[next_share++] : std::min(B, n_kernels - next);
                g->first_eval = next;
                g->step = 0;
                g->active = true;
                ++live;
                for (int e = 0; e < g->cnt; ++e) {
                    const int rec = gs_prof_begin(ctx, g->run, GS_PROF_BUILD, 0.0);
                    const int rc = gs_launch_build<false>(ctx, g->run, g->pool.A + (int64_t)e * g->pool.strideA, ld, ctx->in->X, nullptr, n, n,
                                                          np, np, d, &kernels[next + e], nugget, ctx->build_lower_only);
                    gs_prof_end(ctx, g->run, rec);
                    if (rc) return rc;
                }
                next += g->cnt;
                gs_wave_fill_chain(g, &ca, false);
                const int rec = gs_prof_begin(ctx, g->run, GS_PROF_OTHER, 0.0);
                gs_wv_zsets zs;
                for (int e = 0; e < g->cnt; ++e) zs.off[e] = (int64_t)(gs_z_of(ctx, g->first_eval + e) - ctx->in->Z);
                hipLaunchKernelGGL(k_set_border_g, dim3((unsigned)((naug + 255) / 256), (unsigned)g->cnt), dim3(256), 0, g->run, ca, (int)n,
                                   (const double*)ctx->in->Z, k, zs);
                hipLaunchKernelGGL(k_wave_begin, dim3((unsigned)((np + 255) / 256), (unsigned)g->cnt), dim3(256), 0, g->run, ca);
                gs_prof_end(ctx, g->run, rec);
                GS_CHECK(hipGetLastError());
            } else {
                GS_CHECK(hipStreamWaitEvent(g->run, g->evBulk, 0));            // the trailing update of the previous step
            }
            // ---- one macro-step: the chain of outer step g->step (diagonal super-blocks, then both panels of all rows below them) and
            // its trailing update.  A "near" update (the next panel's 256 columns only, K = 256: ~1 GF per member) sits on the chain's
            // critical path -- chain(s) -> near(s) -> chain(s + 1) -- and goes out on the CHAIN stream, followed at once by the next
            // step's chain; only the big updates (whole lower triangle, K = 512 or 256) go to the bulk stream.  So between two of its
            // big updates a group needs diag + panel + near + diag + panel (~0.8 ms) and the other groups' big updates cover it.
            bool last_on_chain = false;
            for (;;) {
                {
                    const int rec = gs_prof_begin(ctx, g->run, GS_PROF_DIAG, (double)g->cnt * 8.0 * GS_NB * GS_NB * GS_NB / 3.0);
                    gs_wave_fill_chain(g, &ca, false);
                    hipLaunchKernelGGL(k_potrf_diag256g, dim3((unsigned)g->cnt), dim3(256), 0, g->run, ca);
                    gs_prof_end(ctx, g->run, rec);
                }
                const int64_t c0 = 2 * GS_NB * (int64_t)g->step, r2 = c0 + 2 * GS_NB, mrest = naug - r2;
                const bool serial = ctx->wave_serial != 0;
                if (serial) {                  // only the diagonal blocks run beside the bulk stream's kernels (see wave_serial)
                    GS_CHECK(hipEventRecord(g->evChain, g->run));
                    GS_CHECK(hipStreamWaitEvent(wv->sb, g->evChain, 0));
                }
                {
                    hipStream_t spn = serial ? wv->sb : g->run;
                    const int groups = gs_wave_fill_chain(g, &ca, true);
                    const int rec = gs_prof_begin(ctx, spn, GS_PROF_PANEL, (double)g->cnt * 4.0 * (double)mrest * GS_NB * GS_NB);
                    if (ctx->wave_panel_wg4 == 8) hipLaunchKernelGGL(k_panel256gw<8>, dim3((unsigned)((groups + 7) / 8)), dim3(512), 0, spn, ca);
                    else if (ctx->wave_panel_wg4 && ctx->wave_panel_rows_lds) hipLaunchKernelGGL((k_panel256gw<4, true>), dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, spn, ca);
                    else if (ctx->wave_panel_wg4) hipLaunchKernelGGL(k_panel256gw<4>, dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, spn, ca);
                    else hipLaunchKernelGGL(k_panel256g, dim3((unsigned)groups), dim3(64), 0, spn, ca);
                    gs_prof_end(ctx, spn, rec);
                }
                GS_CHECK(hipGetLastError());
                const gs_wave_step st = plans[i][(size_t)g->step];
                // own: the update goes out on the group's CHAIN stream -- the near updates, and (option wave_tail_rows, calls of many rounds)
                // the far updates of a group's last, small macro-steps: on the shared bulk stream a group in its latency-bound last steps is
                // paced by the other groups' 5-ms updates queued in front of its own, which is why groups out of phase lost in round 4
                const bool own_far = !st.near && !serial && ((tail_rows > 0 && mrest <= tail_rows) || ctx->wave_far_own);
                const bool near = (st.near && ctx->wave_near_on_chain && !serial) || own_far;
                last_on_chain = own_far;
                hipStream_t su = near ? g->run : wv->sb;
                if (!near && !serial) {
                    GS_CHECK(hipEventRecord(g->evChain, g->run));
                    GS_CHECK(hipStreamWaitEvent(wv->sb, g->evChain, 0));
                }
                gs_wv_gemm_args ga;
                ga.base = g->pool.A;
                ga.strideA = g->pool.strideA;
                ga.ld = ld;
                ga.n = g->cnt;
                ga.pad = 0;
                gs_wv_gemm_entry en;
                en.offC = r2 * ld + r2;
                en.offA = en.offB = r2 * ld + 2 * GS_NB * (int64_t)st.first;      // panels first .. step: K contiguous columns
                en.M = (int)mrest;
                en.N = st.near ? 2 * GS_NB : (int)mrest;
                en.K = st.K;
                en.tri = st.near ? 0 : 1;
                en.pad = 0;
                const int64_t tm = (mrest + 127) / 128;
                // large lower triangles on the 128 x 128 tile (round 5; alone +2.7 % at M = 15120, +1.5 % at 11280, -7 % on ONE matrix at 7184:
                // profiles/r05_tile128.log), far updates on the bulk stream only
                const bool big_tile = en.tri && !near && ctx->wave_tile128_rows > 0 && mrest >= ctx->wave_tile128_rows;
                const int tiles = (int)(big_tile ? tm * (tm + 1) / 2 : (en.tri ? gs_tri_tiles64(mrest) : tm * ((en.N + 63) / 64)));
                const double fl = en.tri ? (double)mrest * (double)(mrest + 1) * en.K
                                         : (double)en.K * (2.0 * (double)mrest * en.N - (double)en.N * (en.N - 1));
                int run = 0;
                for (int e = 0; e < g->cnt; ++e) {
                    en.q = e;
                    ga.e[e] = en;
                    run += tiles;
                    ga.end[e] = run;
                }
                const int rec = gs_prof_begin(ctx, su, near ? GS_PROF_PANEL : GS_PROF_BULK, fl * g->cnt);     // (near updates on the bulk stream: the same kernel, the same class)
                const size_t shm = 2 * (size_t)((128 + (big_tile ? 128 : 64)) * GS_KC + 4) * sizeof(double);
                if (big_tile && !ctx->lds_attr_done.count((const void*)k_gemm_ld3g2)) {
                    GS_CHECK(hipFuncSetAttribute((const void*)k_gemm_ld3g2, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
                    ctx->lds_attr_done.insert((const void*)k_gemm_ld3g2);
                }
                if (near) hipLaunchKernelGGL(k_gemm_ld3n, dim3((unsigned)run), dim3(512), shm, su, ga);
                else if (big_tile) hipLaunchKernelGGL(k_gemm_ld3g2, dim3((unsigned)run), dim3(512), shm, su, ga);
                else hipLaunchKernelGGL(k_gemm_ld3g, dim3((unsigned)run), dim3(512), shm, su, ga);
                gs_prof_end(ctx, su, rec);
                GS_CHECK(hipGetLastError());
                ++g->step;
                if (!near || own_far) break;
            }
            GS_CHECK(hipEventRecord(g->evBulk, last_on_chain ? g->run : wv->sb));
            if (g->step < S) continue;
            // ---- the round is complete: read-out on the chain stream (the bulk stream goes on with the other groups)
            GS_CHECK(hipStreamWaitEvent(g->run, g->evBulk, 0));
            gs_wave_fill_chain(g, &ca, false);
            const int rec = gs_prof_begin(ctx, g->run, GS_PROF_OTHER, 0.0);
            hipLaunchKernelGGL(k_finalize_g, dim3((unsigned)g->cnt), dim3(256), 0, g->run, ca);
            gs_prof_end(ctx, g->run, rec);
            GS_CHECK(hipGetLastError());
            GS_CHECK(hipMemcpyAsync(ctx->hbatch + (size_t)g->first_eval * 258, g->pool.res, (size_t)g->cnt * 258 * sizeof(double),
                                    hipMemcpyDeviceToHost, g->run));
            g->active = false;
            --live;
        }
    }
    for (int i = 0; i < G; ++i) GS_CHECK(hipStreamSynchronize(wv->g[i].run));
    GS_CHECK(hipStreamSynchronize(wv->sb));
    for (int i = 0; i < n_kernels; ++i) {
        const double* r = ctx->hbatch + (size_t)i * 258;
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) G_out[(size_t)i * k * k + a * k + b] = r[a * 16 + b];
        sld_out[i] = r[256];
        info_out[i] = (int64_t)r[257];
    }
    return 0;
}

