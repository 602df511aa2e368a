// contexts, slots, the groups of a batch, per-launch profile records
// (part of gsum_capi.hip: included from there, in order -- one translation unit)
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <vector>

struct gsum_mat {
    int64_t n = 0, np = 0, ld = 0;
    int T = 0;                 // np / 128
    double* A = nullptr;       // (np + 16) x ld augmented matrix
    double* Linv = nullptr;    // T x 128 x 128 inverses of the diagonal blocks of L: built lazily from the tables, for the one consumer
                               // that multiplies by L_bb^-1 (gsum_cho_solve's back-substitution)
    double* Ltab = nullptr;    // T x GS_LTAB substitution tables of the diagonal blocks
    double* Lsib = nullptr;    // (T / 2 + 1) x GS_LSIB: L(j+1, j) of every outer step in operand layout (k_potrf_diag256 -> k_panel256)
    bool have_ltab = false, have_linv = false, have_lsib = false;   // (have_lsib: Lsib holds the images of THIS factor: gs_need_lsib)
    std::vector<double> solved_rhs;     // host copy of the right-hand sides whose forward solve W^T = (L^-1 RHS)^T the border rows
    int solved_k = -1;                  // hold (-1: none): a repeated predict / forward_gram with the same RHS skips the solve
    double* logdet = nullptr;  // T per-block sums of log L_ii
    double* diag0 = nullptr;   // np original diagonal entries (pivot-cancellation test)
    // persistent-chain schedule (allocated the first time a factorisation of this matrix uses it)
    unsigned* cflags = nullptr;            // gs_fl_count(T / 2) words, zeroed before every factorisation; then T / 2 words "fbwant"
    int fbwant_key = -1;                   // what fbwant was last computed for (window rows x 2 + lazy): uploaded only when it changes
    std::vector<unsigned> fbwant_host;     // ... and its host copy (source of the asynchronous upload)
    double* cdump = nullptr;               // 2 x GS_CH_GMAX x 16 x 256 doubles: operand images of the window's rows
    unsigned long long* cstamps = nullptr; // T / 2 x GS_CH_STAMPS realtime stamps (option "chain_stamps")
    bool factored = false;
};

// One evaluation pipeline: a main + a high-priority panel stream, the events that tie them together, a
// result buffer and a workspace matrix.  Independent evaluations of a batch run on different slots, so
// the latency-bound panel chain of one overlaps the bulk GEMMs of the others.
struct gs_slot {
    hipStream_t sm = nullptr, sp = nullptr;   // main (bulk) / high-priority panel chain
    bool own_su = true;
    bool own_sm = true;              // slots 1-3 (the gradient batch's other evaluations in flight) run on slot 0's other streams
    hipStream_t su = nullptr;        // gradient path: the U = L^-T sweep, trailing the factorisation panel by panel
    hipEvent_t evU = nullptr;
    hipStream_t sa = nullptr;        // persistent-chain schedule: the panel of the rows below the window and the near updates A, B
    hipEvent_t evC = nullptr, evS = nullptr;     // ... its joins (chain kernel / stream sa -> main stream)
    hipEvent_t evN = nullptr;                    // ... and of the near-band stream of the deep schedule (the context's fourth stream)
    std::vector<hipEvent_t> evP, evM, evA;
    hipEvent_t evFork = nullptr;
    hipEvent_t tev[4] = {nullptr, nullptr, nullptr, nullptr};
    double* dres = nullptr; int* dinfo = nullptr;
    double* hres = nullptr;          // pinned
    gsum_mat* ws = nullptr;          // workspace matrix of the fused path (reused across calls)
    int pending = -1;                // index of the evaluation in flight on this slot
    double* gws = nullptr; size_t gws_cap = 0;   // gradient path: this slot's U = L^-T, R^-1, V^T, per-parameter partials
    double* hgrad = nullptr;                     // ... and its pinned read-back buffer (GSUM_MAX_GRAD x 257)
    gsum_kernel_desc last_desc;      // ... and what it was (a chain-schedule timeout re-runs it on the host-enqueued schedule)
    double last_nugget = 0.0;
    int last_index = 0;              // ... and its index in the call (its right-hand-side set)
};

#define GS_MAX_SLOTS 24

// ---- grouped batch schedule ---------------------------------------------------------------------------------------------------------
#define GS_WV_STREAM_GROUPS 4        // groups with a chain stream of their own (option wave_groups)
#define GS_WV_GROUPS 8               // ... times two COHORTS in calls of many rounds: group i + G runs on group i's stream, half a round behind
struct gs_wave_group {
    hipStream_t sc = nullptr;            // this group's chain stream (high priority): kernel builds, diagonal blocks, panels, read-out
    hipStream_t run = nullptr;           // ... the stream it runs on in the current call: sc, or -- a second cohort -- the stream of the group it shadows
    bool own_sc = false;                 // groups 0 and 1 borrow slot 0's panel and auxiliary streams (see gs_wave_prepare)
    hipEvent_t evChain = nullptr, evBulk = nullptr;
    gs_wv_pool pool;                     // `cap` workspaces at fixed strides
    int cap = 0;
    int64_t n = 0;                       // order the pool was allocated for
    // state inside a call
    int cnt = 0, step = 0, first_eval = 0, start_tick = 0;
    bool active = false;
    gs_wave_group() { memset(&pool, 0, sizeof pool); }
};
struct gs_wave {
    hipStream_t sb = nullptr;            // the bulk stream: the trailing updates of all groups, one launch after the other (slot 0's main stream)
    gs_wave_group g[GS_WV_GROUPS];
};

struct gs_inputs {
    double* X = nullptr; int64_t n = 0; int d = 0; size_t X_cap = 0;     // n x d points
    double* Z = nullptr; int k = 0; size_t Z_cap = 0;                    // n x k right-hand sides; n_sets of them back to back
    int n_sets = 1;                                                      // (gsum_set_inputs_sets: evaluation i of a call reads set set_of[i])
    std::vector<double> x_host, z_host;                                  // host copies of SMALL inputs (gs_upload_X / gs_upload_Z skip an equal upload)
    int64_t z_rows = 0;
};

// state of a gradient evaluation's stage behind the factorisation (api_grad.hip.h), kept in the context so that gs_potrf_chain's step hook can
// enqueue the sweep's launches between the factorisation's
struct gs_grad_run {
    bool prepared = false;
    int next_c = 0;                      // first block column of the sweep not yet enqueued
    gs_slot* sl = nullptr;
    gsum_mat* m = nullptr;
    int P = 0, d = 0, chunks = 0, rows_per = 0;
    bool solo = false, on_chain = false, trail = false, have_sib = false, lazy_ok = false, deferred = false;
    int64_t n = 0, np = 0, ld = 0, ldg = 0;
    double *U = nullptr, *Ri = nullptr, *Vt = nullptr, *Q = nullptr, *trow = nullptr, *dout = nullptr, *part = nullptr;
    bool split = false;                  // Q_p beside the R^-1 product, traces from stored triangles of dR_p (gs_grad_post)
    double* dR = nullptr;
    int64_t dr_stride = 0;
    hipStream_t s = nullptr, su = nullptr;
};

struct gsum_ctx {
    int device = 0;
    gs_slot slots[GS_MAX_SLOTS];
    int n_slots_ready = 0;
    gs_slot* cur = nullptr;          // slot the helpers below enqueue on
    bool upload_pending = false;     // a host -> device copy of inputs was enqueued since the last synchronisation of the main stream
    int64_t uploads_skipped = 0;     // uploads of small inputs whose bytes were already on the device
    int grad_batch_wave = 1;         // gradient batches: the factorisations on the grouped schedule (gs_grad_batch_wave)
    int grad_interleave = 1;         // one gradient evaluation alone: the U = L^-T sweep's launches enqueued step by step between the factorisation's
    int grad_split = 1;              // ... and its kernel-gradient contractions split: Q_p beside the R^-1 product, the traces from stored triangles of dR_p
    int grad_lazy_chain = 1;         // ... its trailing updates paired (K = 512 every other pair) also when it trails the persistent chain by flags
    int (*chain_step_hook)(gsum_ctx*, gsum_mat*, int) = nullptr;      // gs_potrf_chain calls it when outer step s is enqueued (the gradient path's sweep)
    int chain_hook_P = 0;
    gs_grad_run grad_run;
    int batch_slots = 4;             // gradient evaluations kept in flight by gsum_lml_grad_batch, one stream each: the context's four
                                     // streams on four pipes (n = 8192: 14.3 / 13.3 / 12.4 / 12.2 / 12.4 ms each with 2 / 3 / 4 / 5 / 8;
                                     // value-only batches do not use slots: gs_lml_wave)
    int batch_active = 1;            // evaluations in flight in the current call (look-ahead is used only alone)
    int prio_lo = 0, prio_hi = 0;
    std::string err;
    int lookahead = 1;
    double next_algo_flops = -1.0;   // profile only: algorithmic flops of the next cfg-5 launch when not M(M+1)K / 2MNK
    int predict_lookahead = 1;       // the predictive sweep with its panels and near updates on the chain stream, one far launch per macro-step beside them
    int predict_depth = 2;           // pairs of block columns per macro-step of the predictive sweep (2: the pairing of round 3)
    int predict_split = 0;           // ... in two independent half-sweeps (rows of the new points) on two streams from 1024 rows up
    int predict_panel256 = 1;        // the predictive sweep's panel pairs as ONE k_panel256 launch (0: k_panel, K = 128 GEMM, k_panel)
    int predict_lazy = 1;            // the predictive sweep V^T = K* L^-T with the same pairing of trailing updates (K = 512 every other step)
    int lazy_min_np = 4352;          // smallest padded order the lazy far updates are used at (profiles/r03_lazy_threshold.log, 20 in flight: +3 % at 4352,
                                     // +3.7 / +4.8 / +5.3 / +6 / +6 % at 5120 / 6144 / 7168 / 8192 / 12288; neutral at 4096, -0.5 ... -3 % at 1536 ... 3072)
    int lazy_far = 2;                // batch mode: K = 512 updates of the far trailing region every other panel (1: the next TWO panels' columns are "near",
                                     // updated with K = 256 at every step; 2: only the next panel's, the one after it takes both updates in the K = 512 launch)
    int bench_fill = 0;              // gsum_bench_gemm_nt operands: 0 random, 1 zeros (timing is value-independent, board power is not)
    int build_lower_only = 1;
    int bulk_lds_pad = 80 * 1024;    // bytes of dynamic LDS the bulk kernel asks for in the look-ahead schedule of a factorisation
                                     // (0 = what it needs, 53 KB): at 80 KB two bulk workgroups share a CU instead of three and a
                                     // retiring one leaves room for a chain workgroup at once -- one factorisation 6.85 -> 6.70 ms
    bool bulk_pad_now = false;       // set around the bulk launches of gs_potrf's look-ahead branch only
    int chain_prefetch = 1;          // 32 x 128 tile (sibling / look-ahead updates): four operand chunks in flight instead of one
    int la_depth2 = 1;               // look-ahead schedule: the bulk update in two launches, the chain waits for the first only
                                     // (-1 % with the 80-KB bulk launches: 6.69 -> 6.62 ms; nothing without them)
    int chain_fused = -1;            // two diagonal blocks per launch (k_potrf_diag256) and both panels of the rows below in one
                                     // (k_panel256) instead of diag / panel / sibling update / diag / panel: 1 = always, 0 = never,
                                     // -1 (default) = in batches only.  The fused kernels are slower end to end (125 + 35 us against
                                     // 31 + 12 + 11 + 31 + 12) but two launches instead of five and less CU time: with 16 evaluations
                                     // in flight latency is hidden and the batch runs 1.8 % faster (279 vs 274 evals/s), one
                                     // factorisation alone is 10-30 % slower with them
    int chain_persist = -1;          // ONE factorisation alone: the dependent chain as a persistent kernel on CUs of its own (k_chain),
                                     // the M-proportional work host-enqueued and gated on its flags.  -1 (default) = when the order
                                     // is a multiple of 256 and at least chain_min_np, 1 = whenever the order allows, 0 = never
    int chain_min_np = 768;           // (round 4: 2048 -> 768; n = 768 ... 1536: 9-14 % shorter, bit-identical)
    int chain_lazy = -1;              // persistent-chain schedule: far region of the trailing matrix updated every other step with K = 512 (measured: no gain at n = 8192 -- the K = 512 launch reaches 47 TF/s in situ, not the 55 of the microbenchmark, and the near-only steps leave the chip half empty; +1 % at 4096)
                                     // -1 (default): on from padded order 10240 up, where it pays -- 13.6 -> 13.3 ms at n = 12288, 29.15 -> 28.13 ms at 16384, 5.28 -> 5.31 at 8192
    int chain_rows = 512;            // the chain's window: rows under the panel it solves and updates itself (256 or 512)
    int chain_stamps = 0;            // record the chain kernel's per-step realtime stamps (gsum_debug_chain_stamps)
    int chain_probe = 0;             // two-stream concurrency probe: 0 not run, 1 streams run side by side, -1 they do not (a
                                     // profiler serialises dispatches): the chain schedule would deadlock until its timeout
    int pipes_ok = -1;               // pairwise stream probe of gsum_init (gs_pipe_probe): 1 every pair of the context's four streams ran side by
                                     // side, 0 some pair took turns (two streams on one command-processor pipe, or a tool that serialises
                                     // dispatches: a batch loses 3-6 %, a single factorisation 30-70 %), -1 not run
    int pipe_heals = 0;              // streams gsum_init replaced because they took turns with another of the four
    int pipe_overlap_permille = -1;  // ... the smallest overlap of a pair's two 100-us kernels, in 1/1000 of their length
    int chain_events_needed = 0;     // 1: a caller trails the factorisation by evP events (host-enqueued schedule only); 2: the gradient
                                     // path: either schedule, it trails the persistent chain by the chain's own flags
    bool last_potrf_chain = false;   // the last gs_potrf took the persistent-chain schedule
    int chain_aborts = 0;            // factorisations whose chain kernel timed out (the schedule is then switched off)
    int chain_test_abort = 0;        // test hook: the chain gives up at this outer step of its NEXT factorisation (one shot)
    unsigned long long* kst_ptr = nullptr;   // diagnostics: start / end stamp pair of the NEXT bulk (cfg 7) / k_panel256 launch
    unsigned long long* panel_stats = nullptr;   // diagnostics (option panel_stats): {sum of wave lifetimes in 10-ns ticks, waves} of every k_panel256 launch
    int first_tiles = 0;                  // the NEXT bulk (cfg 7) launch: its first-256-column tiles first, counted in *first_done (k_gemm_ld3)
    unsigned* first_done = nullptr;
    int second_c2 = 0;                    // ... and a second counted group behind them: its column tiles 4 .. second_c2 - 1, counted in *second_done
    unsigned* second_done = nullptr;
    int chain_deep = -1;                  // persistent-chain schedule: trailing updates grouped `chain_depth` panels deep while the far region is large
                                          // (gs_potrf_chain: near band on the context's fourth stream, ONE K = 256 x depth far launch per macro-step);
                                          // -1 (default): from padded order 10240 up, 0 never, 1 whenever a macro-step fits.  Measured
                                          // (profiles/r05_chain_deep.log): n = 16384 27.70 -> 27.07 ms, 12288 13.12 -> 12.88, 8192 5.23 -> 5.43-5.51
                                          // (depth 2: 5.18), 4096 unchanged -- the head is bound by the chip's aggregate rate, not by waits: the
                                          // K = 1024 launches run at 53 TF/s beside the near launches (65 alone) and those at K = 256
    int chain_depth = 4;
    int chain_deep_rows = 4096;           // ... as long as the far region of the macro-step has at least this many rows
    // Inputs on the device.  `res` is written by gsum_set_inputs ONLY and read by gsum_lml_resident; every other entry
    // point (operator level, gsum_lml_batch, gsum_lml_grad) uploads into `op`.  `in` is the set the fused path reads.
    gs_inputs op, res;
    gs_inputs* in = &res;
    const int32_t* set_of = nullptr;   // the current call's right-hand-side set per evaluation (host array; NULL: set 0 throughout)
    double* scratch = nullptr; size_t scratch_cap = 0;
    double* hbatch = nullptr; size_t hbatch_cap = 0;   // pinned host buffer for the fused paths' result blocks (258 doubles each)
    double* gws = nullptr; size_t gws_cap = 0;     // gradient path: U = L^-T, R^-1, V^T, per-parameter partials
    double timers[4] = {0, 0, 0, 0};
    unsigned long long* dstamps = nullptr;   // 8 u64: phase stamps of the last diagonal-block kernel
    int diag_stamps = 0;
    // optional per-launch HIP-event profile of the big-tile (cfg 0) GEMM launches
    int profile_gemm = 0;            // N > 0: HIP events around the bulk launches of every N-th fused evaluation
    int prof_eval_count = 0;         // fused evaluations enqueued since profiling was switched on
    bool prof_this_eval = true;
    std::vector<hipEvent_t> prof_pool;
    struct ProfRec { int e0, e1; double flops; int cls; };     // cls: GS_PROF_* below
    std::vector<ProfRec> prof_recs;
    size_t prof_next = 0;
    int small_path = 1;              // n <= 128: fused one-workgroup-per-evaluation kernel
    int medium_path = 1;             // 128 < n <= 2048 and >= medium_min_batch evaluations per call: one workgroup per
    int medium_min_batch = -1;       // evaluation on its own HBM-resident matrix (k_lml_medium); -1 = auto: max(4, n^1.45 / 985),
                                     // the measured break-even against the pipelined multi-kernel path
    double host_enqueue_ms = 0.0;    // host wall time spent enqueuing the last evaluation
    std::set<const void*> lds_attr_done;   // kernels whose dynamic-LDS limit has been raised on this context's device
    // grouped batch schedule (gs_lml_wave): the evaluations of a call advance in groups, one launch per kernel class and outer step
    gs_wave wave;
    int wave_groups = 3;             // groups = chain streams; their bulk launches alternate on ONE bulk stream (4 streams: the HIP runtime's
                                     // default number of hardware queues)
    int wave_size = 8;               // evaluations per group at most
    int wave_tile128_rows = 0;       // far updates of at least this many rows on the 128 x 128 workgroup tile (k_gemm_ld3g2); 0 (default): never.  Alone the
                                     // tile is +2.7 % at M = 15120 and +1.5 % at 11280; inside a batch it LOSES (profiles/r05_tile128.log: n = 16384 45.4
                                     // against 45.6 evals/s, n = 12288 104.6 / 105.3, n = 8192 from 6144 rows 327.8 / 333.7): two 66-KB workgroups of
                                     // 380-us tiles per CU leave the chain kernels less room than three 48-KB ones of 190 us
    int wave_far_own = 0;            // (lab) every far update on its group's chain stream: the groups' far launches then overlap one another's ramp and
                                     // drain, and the chain kernels lose their priority over them: 336.6 against 339.3 evals/s (20 per call), 345.8 / 350.4
                                     // (96 per call); profiles/r05_far_own.log -- the one low-priority bulk stream stays
    int wave_cohorts = 2;            // calls of at least wave_cohort_min x (groups x size) evaluations: every group runs TWO cohorts of evaluations half a
    int wave_cohort_min = 4;         // round apart on its one chain stream, so that the latency-bound last steps of one cohort run under the far updates
                                     // of the other (gs_lml_wave); 1: one cohort (rounds in phase, each ending with ~2.5 ms of latency-bound steps)
    int wave_long_rounds = 6;        // a call of at least this many rounds is "long": groups out of phase (wave_shift = -1) and own-stream tails
    int wave_tail_rows = 2048;       // with two cohorts: far updates of at most this many rows go out on the group's chain stream, not on the shared bulk
                                     // stream, where a cohort in its latency-bound last steps would queue behind the other groups' 5-ms updates
                                     // (< 0: in every call of several rounds; 0: never).  n = 8192, 1536 evaluations per call
                                     // (profiles/r05_long_call.log): one cohort 337.0 evals/s, two 339.1, two + own-stream tails 343.8 (345.9 with 12 per cohort)
    int wave_shift = 0;              // macro-steps by which consecutive groups are out of phase in calls of several rounds (0: in phase)
    int wave_panel_rows_lds = 1;     // ... their rows go global <-> registers as whole 128-B lines and change layout in LDS
    int wave_head = 124;               // first macro-step lengths of the groups in a call (decimal digits; 0: all `wave_depth`)
    int wave_min = 3;                // calls with at least this many evaluations take the grouped schedule
    int wave_last_streams = 0;
    int wave_panel_wg4 = 4;          // waves per workgroup of a batch's panel solves (k_panel256gw): 0 = one (k_panel256g), 4 (default), 8.
                                     // n = 8192, 3 groups of 7 (tools/gpu_wave_profile.py): 321 / 325 / 314 evals/s with 1 / 4 / 8 waves per
                                     // workgroup at 20 evaluations per call, 323.5 / 327.7 / 316.8 at 84.  With 8 the panel waves own whole
                                     // CUs and the other groups' far updates run at 62-64 TF/s instead of 55 -- but the panels take 2.5 x longer
                                     // (they wait for CUs to empty) and become the critical path
    int wave_serial = 0;             // 1: a group's panels and ALL its trailing updates on the one bulk stream, only its diagonal blocks on the
                                     // chain stream.  Panels and near updates are chip-filling MFMA work themselves (7 + 5.5 ms of a 20-evaluation
                                     // call at n = 8192, against 48 ms of far updates): run beside the far updates of another group they
                                     // slow those down by as much as they take (far updates 65.8 TF/s alone, 56.5 beside them), so nothing
                                     // is gained by the overlap and no per-launch time means anything.  One after the other every kernel
                                     // runs at its exclusive rate and the sum of the bulk stream's launches IS the step time; what still
                                     // overlaps is what is latency-bound: the diagonal blocks (10 workgroups) and the kernel builds.
                                     // 0 (default): panels and near updates on the chain streams.  Measured (tools/gpu_wave_profile.py,
                                     // 2 x 10): serial 300 evals/s with the bulk stream's launches at 63.5 TF/s, overlapped 316 with
                                     // the far updates at 56.5 -- the overlap does hide ~4 ms of a 64-ms call (launch gaps, the
                                     // panels' latency-bound share), so it stays the default
    int wave_near_on_chain = 1;      // the small "near" trailing updates (K = 256, the next panel's columns only) on the group's chain stream
    int wave_depth = 4;              // panels per macro-step of the batch schedule: the far trailing region is updated once per `wave_depth`
                                     // panels with K = 256 x wave_depth (2: the pairing of rounds 2-3).  n = 8192, 20 evaluations per call
                                     // (tools/gpu_wave_check.py, 2 groups of 10): 313.0 / 316.4 / 315.9 / 314.7 / 312.0 evals/s at depth 2 / 3 / 4 / 6 / 8;
                                     // 3 groups of 7 (tools/gpu_wave_profile.py): 315.3 / 320.4 / 321.3 at depth 2 / 3 / 4, 84 per call 316.3 / 321.4 / 322.2
    int wave_deep_rows = 3072;       // ... deeper than 2 only while the trailing matrix has at least this many rows
};

static std::string g_init_error;
static void gs_wave_release(gsum_ctx* ctx, bool streams);

// kernel classes of the per-launch HIP-event profile (option "profile_gemm")
enum { GS_PROF_BUILD = 0, GS_PROF_DIAG = 1, GS_PROF_PANEL = 2, GS_PROF_BULK = 3, GS_PROF_OTHER = 4, GS_PROF_CLASSES = 5 };

#define GS_CHECK(expr)                                                                             \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            char buf_[512];                                                                        \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            ctx->err = buf_;                                                                       \
            return -1;                                                                             \
        }                                                                                          \
    } while (0)

#define GS_FAIL(msg)            \
    do {                        \
        ctx->err = (msg);       \
        return -2;              \
    } while (0)

static int gs_reserve(gsum_ctx* ctx, double** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes && *p) return 0;
    if (*p) GS_CHECK(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    GS_CHECK(hipMalloc((void**)p, bytes));
    *cap = bytes;
    return 0;
}

// Bracket the launches enqueued between begin and end (one kernel, as a rule) with HIP events on THEIR stream; only
// while an evaluation is being profiled.  Returns the record index to hand to gs_prof_end, or -1.
static int gs_prof_begin(gsum_ctx* ctx, hipStream_t s, int cls, double flops) {
    if (!ctx->profile_gemm || !ctx->prof_this_eval) return -1;
    while (ctx->prof_pool.size() < ctx->prof_next + 2) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return -1;
        ctx->prof_pool.push_back(ev);
    }
    const int e0 = (int)ctx->prof_next, e1 = e0 + 1;
    ctx->prof_next += 2;
    if (hipEventRecord(ctx->prof_pool[e0], s) != hipSuccess) return -1;
    ctx->prof_recs.push_back({e0, e1, flops, cls});
    return (int)ctx->prof_recs.size() - 1;
}

static void gs_prof_end(gsum_ctx* ctx, hipStream_t s, int rec) {
    if (rec >= 0) (void)hipEventRecord(ctx->prof_pool[ctx->prof_recs[rec].e1], s);
}

