// libgsum_hip.so — host side of the C ABI declared in include/gsum_hip.h.
// One context = one GPU = two HIP streams (main + high-priority panel stream for look-ahead).
// the library is built with -fvisibility=hidden: what include/gsum_hip.h declares (and, in the lab build, gsum_hip_debug.h) is all it exports
#pragma GCC visibility push(default)
#include "gsum_hip.h"
#ifdef GSUM_LAB
#include "gsum_hip_debug.h"
#endif
#pragma GCC visibility pop
#include "gsum_kernels.hip.h"

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
    bool have_ltab = false, have_linv = false;
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
};

#define GS_MAX_SLOTS 24

// ---- grouped batch schedule ---------------------------------------------------------------------------------------------------------
#define GS_WV_GROUPS 4
struct gs_wave_group {
    hipStream_t sc = nullptr;            // this group's chain stream (high priority): kernel builds, diagonal blocks, panels, read-out
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
    double* Z = nullptr; int k = 0; size_t Z_cap = 0;                    // n x k right-hand sides
};

struct gsum_ctx {
    int device = 0;
    gs_slot slots[GS_MAX_SLOTS];
    int n_slots_ready = 0;
    gs_slot* cur = nullptr;          // slot the helpers below enqueue on
    int batch_slots = 4;             // gradient evaluations kept in flight by gsum_lml_grad_batch, one stream each: the context's four
                                     // streams on four pipes (n = 8192: 14.3 / 13.3 / 12.4 / 12.2 / 12.4 ms each with 2 / 3 / 4 / 5 / 8;
                                     // value-only batches do not use slots: gs_lml_wave)
    int batch_active = 1;            // evaluations in flight in the current call (look-ahead is used only alone)
    int prio_lo = 0, prio_hi = 0;
    std::string err;
    int lookahead = 1;
    double next_algo_flops = -1.0;   // profile only: algorithmic flops of the next cfg-5 launch when not M(M+1)K / 2MNK
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
    int chain_events_needed = 0;     // the gradient path trails the factorisation by evP events: host-enqueued schedule only
    int chain_aborts = 0;            // factorisations whose chain kernel timed out (the schedule is then switched off)
    int chain_test_abort = 0;        // test hook: the chain gives up at this outer step of its NEXT factorisation (one shot)
    unsigned long long* kst_ptr = nullptr;   // diagnostics: start / end stamp pair of the NEXT bulk (cfg 7) / k_panel256 launch
    unsigned long long* panel_stats = nullptr;   // diagnostics (option panel_stats): {sum of wave lifetimes in 10-ns ticks, waves} of every k_panel256 launch
    int first_tiles = 0;                  // the NEXT bulk (cfg 7) launch: its first-256-column tiles first, counted in *first_done (k_gemm_ld3)
    unsigned* first_done = nullptr;
    // Inputs on the device.  `res` is written by gsum_set_inputs ONLY and read by gsum_lml_resident; every other entry
    // point (operator level, gsum_lml_batch, gsum_lml_grad) uploads into `op`.  `in` is the set the fused path reads.
    gs_inputs op, res;
    gs_inputs* in = &res;
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
            blocks = Tt * (Tt + 1);
        } else {
            blocks = ((M + 127) / 128) * ((N + 63) / 64);
        }
        hipLaunchKernelGGL(k_gemm_ld3<2>, dim3((unsigned)blocks), dim3(512), shmem, s, C, ldc, A, lda, B, ldb, (int)M, (int)N, K, tri,
                           beta, sign, ctx->kst_ptr, tri == 2 ? 0 : ctx->first_tiles, ctx->first_done);
        ctx->kst_ptr = nullptr;
        ctx->first_tiles = 0;
        ctx->first_done = nullptr;
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

// ---- matrices ---------------------------------------------------------------------------------
// bulk update of a look-ahead schedule: the one launch class that asks for `bulk_lds_pad` bytes of LDS (two workgroups per
// CU, so that chain workgroups find room as soon as one retires); every other user of the bulk tile wants three per CU
static int gs_bulk_la(gsum_ctx* ctx, hipStream_t s, int cfg, double* C, int64_t ldc, const double* A, int64_t lda, const double* B,
                      int64_t ldb, int64_t M, int64_t N, int K, int tri, int beta, double sign) {
    ctx->bulk_pad_now = true;
    const int rc = gs_gemm(ctx, s, cfg, C, ldc, A, lda, B, ldb, M, N, K, tri, beta, sign);
    ctx->bulk_pad_now = false;
    return rc;
}

// Padded order of a matrix (identity padding: exact zeros in every product, so results do not depend on it).  128 is the block
// size; where the persistent-chain schedule applies the order goes to the next multiple of 256 instead -- an even number of block
// columns -- so that schedule serves every order, not only multiples of 256 (n = 7976: 8192 instead of 8064 rows, +1.6 % work for a
// factorisation that is 15 % faster).
static int64_t gs_padded_order(const gsum_ctx* ctx, int64_t n) {
    const int64_t p128 = (n + GS_NB - 1) / GS_NB * GS_NB, p256 = (n + 2 * GS_NB - 1) / (2 * GS_NB) * (2 * GS_NB);
    if (ctx->chain_persist != 0 && p256 >= ctx->chain_min_np) return p256;
    return p128;
}

static int gs_mat_alloc(gsum_ctx* ctx, int64_t n, gsum_mat** out) {
    if (n <= 0 || n > (1 << 20)) GS_FAIL("matrix order out of range");
    gsum_mat* m = new gsum_mat();
    m->n = n;
    m->np = gs_padded_order(ctx, n);
    m->ld = m->np + GS_BORDER;          // row stride 64 KiB + 128 B at n = 8192: no channel aliasing
    m->T = (int)(m->np / GS_NB);
    hipError_t e = hipMalloc((void**)&m->A, (size_t)(m->np + GS_BORDER) * m->ld * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&m->Linv, (size_t)m->T * GS_NB * GS_NB * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&m->Ltab, (size_t)m->T * GS_LTAB * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&m->Lsib, (size_t)(m->T / 2 + 1) * GS_LSIB * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&m->logdet, (size_t)m->T * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&m->diag0, (size_t)m->np * sizeof(double));
    if (e != hipSuccess) {
        for (double** q : {&m->A, &m->Linv, &m->Ltab, &m->Lsib, &m->logdet, &m->diag0}) {
            if (*q) (void)hipFree(*q);
            *q = nullptr;
        }
        delete m;
        ctx->err = std::string("hipMalloc(matrix) failed: ") + hipGetErrorString(e);
        return -1;
    }
    *out = m;
    return 0;
}

static void gs_mat_release(gsum_mat* m) {
    if (!m) return;
    if (m->A) (void)hipFree(m->A);
    if (m->Linv) (void)hipFree(m->Linv);
    if (m->Ltab) (void)hipFree(m->Ltab);
    if (m->Lsib) (void)hipFree(m->Lsib);
    if (m->logdet) (void)hipFree(m->logdet);
    if (m->diag0) (void)hipFree(m->diag0);
    if (m->cflags) (void)hipFree(m->cflags);
    if (m->cdump) (void)hipFree(m->cdump);
    if (m->cstamps) (void)hipFree(m->cstamps);
    delete m;
}

static int gs_check_desc(gsum_ctx* ctx, const gsum_kernel_desc* desc, int d) {
    if (!desc) GS_FAIL("kernel descriptor is NULL");
    if (d < 1 || d > GSUM_MAX_D) GS_FAIL("input dimension must be 1..GSUM_MAX_D");
    if (desc->n_ops == 0) {
        if (desc->family < GSUM_RBF || desc->family > GSUM_MATERN12) GS_FAIL("unknown kernel family");
        int nls = desc->anisotropic ? d : 1;
        for (int i = 0; i < nls; ++i)
            if (!(desc->length_scale[i] > 0.0)) GS_FAIL("length_scale must be positive");
        return 0;
    }
    // a tree: a well-formed postfix program over valid leaves
    if (desc->n_ops < 0 || desc->n_ops > GSUM_MAX_OPS || desc->n_leaves < 1 || desc->n_leaves > GSUM_MAX_LEAVES) GS_FAIL("kernel tree: bad op / leaf count");
    int depth = 0;
    for (int k = 0; k < desc->n_ops; ++k) {
        const int op = desc->op[k];
        if (op == GSUM_OP_ADD || op == GSUM_OP_MUL) {
            if (depth < 2) GS_FAIL("kernel tree: operator without two operands");
            --depth;
        } else {
            const int idx = op >= GSUM_OP_WHITE ? op - GSUM_OP_WHITE : (op >= GSUM_OP_CONST ? op - GSUM_OP_CONST : op - GSUM_OP_LEAF);
            if (op < GSUM_OP_LEAF || idx < 0 || idx >= (op >= GSUM_OP_CONST ? GSUM_MAX_OPS : desc->n_leaves)) GS_FAIL("kernel tree: bad operand");
            if (++depth > 8) GS_FAIL("kernel tree: deeper than 8 pending operands");
        }
    }
    if (depth != 1) GS_FAIL("kernel tree: the program does not reduce to one value");
    for (int l = 0; l < desc->n_leaves; ++l) {
        const gsum_kernel_leaf& lf = desc->leaf[l];
        if (lf.family < GSUM_RBF || lf.family > GSUM_RQ) GS_FAIL("kernel tree: unknown leaf family");
        if (lf.family == GSUM_RQ && (lf.anisotropic || !(lf.alpha > 0.0))) GS_FAIL("kernel tree: RationalQuadratic needs alpha > 0 and an isotropic length scale");
        for (int i = 0; i < (lf.anisotropic ? d : 1); ++i)
            if (!(lf.length_scale[i] > 0.0)) GS_FAIL("length_scale must be positive");
    }
    return 0;
}

// Kernel-matrix build launcher: picks the template instance (family, one-dimensional fast path) of k_build2.
// tri != 0: lower 128-column tiles of a square padded matrix only.
template <bool CROSS>
static int gs_launch_build(gsum_ctx* ctx, hipStream_t s, double* out, int64_t ldo, const double* X, const double* Y, int64_t n,
                           int64_t m, int64_t prow, int64_t pcol, int d, const gsum_kernel_desc* desc, double diag_add, int tri) {
    const int64_t tr = (prow + GS_B2_ROWS - 1) / GS_B2_ROWS, tc = (pcol + 127) / 128, t128 = (prow + 127) / 128;
    const int64_t blocks = tri ? 4 * (t128 * (t128 + 1) / 2) : tr * tc;
    if (desc->n_ops > 0) {                      // a general Sum / Product tree: entry-by-entry evaluation (k_build_tree)
        hipLaunchKernelGGL((k_build_tree<CROSS>), dim3((unsigned)blocks), dim3(256), 0, s, out, ldo, X, Y, (int)n, (int)m, (int)prow,
                           (int)pcol, d, *desc, diag_add, tri);
        GS_CHECK(hipGetLastError());
        return 0;
    }
#define GS_B2_LAUNCH(FAM, D1)                                                                                              \
    hipLaunchKernelGGL((k_build2<CROSS, FAM, D1>), dim3((unsigned)blocks), dim3(256), 0, s, out, ldo, X, Y, (int)n, (int)m, \
                       (int)prow, (int)pcol, d, *desc, diag_add, tri)
    const bool d1 = d == 1;
    switch (desc->family) {
        case GSUM_RBF: if (d1) GS_B2_LAUNCH(GSUM_RBF, true); else GS_B2_LAUNCH(GSUM_RBF, false); break;
        case GSUM_MATERN52: if (d1) GS_B2_LAUNCH(GSUM_MATERN52, true); else GS_B2_LAUNCH(GSUM_MATERN52, false); break;
        case GSUM_MATERN32: if (d1) GS_B2_LAUNCH(GSUM_MATERN32, true); else GS_B2_LAUNCH(GSUM_MATERN32, false); break;
        default: if (d1) GS_B2_LAUNCH(GSUM_MATERN12, true); else GS_B2_LAUNCH(GSUM_MATERN12, false); break;
    }
#undef GS_B2_LAUNCH
    GS_CHECK(hipGetLastError());
    return 0;
}

// K1 into an augmented matrix (square, symmetric form).  X must already be on the device.
static int gs_build_into(gsum_ctx* ctx, hipStream_t s, gsum_mat* m, const gsum_kernel_desc* desc, const double* dX,
                         int d, double diag_add, int lower_only) {
    const int rec = gs_prof_begin(ctx, s, GS_PROF_BUILD, 0.0);
    const int rc = gs_launch_build<false>(ctx, s, m->A, m->ld, dX, nullptr, m->n, m->n, m->np, m->np, d, desc, diag_add, lower_only);
    gs_prof_end(ctx, s, rec);
    if (rc) return rc;
    m->factored = false;
    return 0;
}

static int gs_set_border(gsum_ctx* ctx, hipStream_t s, gsum_mat* m, const double* dZ, int k) {
    int64_t cols = m->np + GS_BORDER;
    const int rec = gs_prof_begin(ctx, s, GS_PROF_OTHER, 0.0);
    hipLaunchKernelGGL(k_set_border, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s, m->A, m->ld, (int)m->n,
                       (int)m->np, dZ, k);
    gs_prof_end(ctx, s, rec);
    GS_CHECK(hipGetLastError());
    return 0;
}

// ---- K2: two-level blocked right-looking Cholesky with look-ahead --------------------------------
// Outer step = a 256-column panel made of two 128-column sub-steps (block columns b, b+1):
//   diag(b)   : L_bb, L_bb^-1                                  (k_potrf_diag, one workgroup)
//   trsm(b)   : rows below  <-  rows * L_bb^-T                 (MFMA GEMM against the explicit inverse)
//   col(b+1)  : block column b+1 -= P_b P_b[b+1]^T             (K = 128, only 128 columns wide)
//   diag(b+1), trsm(b+1)
//   la        : next panel's 256 columns -= P P[next]^T        (K = 256)   } P = both sub-panels,
//   bulk      : everything right of it  -= P P^T, lower tiles  (K = 256)   } 256 contiguous columns
// The trailing matrix is read and written once per 256 eliminated columns: K = 256 doubles the flops
// per byte of C traffic over a plain nb = 128 sweep (the K = 128 update was memory-side bound).  With
// look-ahead, everything but `bulk` runs on the high-priority panel stream, so the panel chain of step
// s+1 overlaps bulk(s).  Both streams are joined on the main stream at the end.
static int gs_potrf_events(gsum_ctx* ctx, gs_slot* sl, int T) {
    if ((int)sl->evP.size() < T + 1) {
        size_t old = sl->evP.size();
        sl->evP.resize(T + 1);
        sl->evM.resize(T + 1);
        sl->evA.resize(T + 1);
        for (size_t i = old; i < sl->evP.size(); ++i) {
            GS_CHECK(hipEventCreateWithFlags(&sl->evP[i], hipEventDisableTiming));
            GS_CHECK(hipEventCreateWithFlags(&sl->evM[i], hipEventDisableTiming));
            GS_CHECK(hipEventCreateWithFlags(&sl->evA[i], hipEventDisableTiming));
        }
    }
    return 0;
}

// The high-priority panel stream exists only on slots that run a look-ahead schedule.
static int gs_panel_stream(gsum_ctx* ctx, gs_slot* sl) {
    if (!sl->sp) GS_CHECK(hipStreamCreateWithPriority(&sl->sp, hipStreamNonBlocking, ctx->prio_hi));
    return 0;
}

static int gs_diag(gsum_ctx* ctx, hipStream_t s, gsum_mat* m, int b) {
    gs_slot* sl = ctx->cur;
    const int64_t c = (int64_t)b * GS_NB;
    unsigned long long* stamps = ctx->diag_stamps ? ctx->dstamps : (unsigned long long*)nullptr;
    const int rec = gs_prof_begin(ctx, s, GS_PROF_DIAG, (double)GS_NB * GS_NB * GS_NB / 3.0);
    hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(256), 0, s, m->A + c * m->ld + c, m->ld, m->Ltab + (size_t)b * GS_LTAB, m->logdet + b,
                       sl->dinfo, (int)c, m->diag0 + c, stamps);
    gs_prof_end(ctx, s, rec);
    GS_CHECK(hipGetLastError());
    return 0;
}

// P (M rows x 128 columns, leading dimension ldp)  <-  P L_bb^-T for diagonal block b of the factor m: blocked
// substitution against the block's tables (k_panel).
static int gs_trsm_rows(gsum_ctx* ctx, hipStream_t s, const gsum_mat* m, int b, double* P, int64_t ldp, int64_t M) {
    if (M <= 0) return 0;
    const int rec = gs_prof_begin(ctx, s, GS_PROF_PANEL, (double)M * GS_NB * GS_NB);
    hipLaunchKernelGGL(k_panel, dim3((unsigned)((M + 15) / 16)), dim3(64), 0, s, P, ldp, (int)M, m->Ltab + (size_t)b * GS_LTAB);
    gs_prof_end(ctx, s, rec);
    GS_CHECK(hipGetLastError());
    return 0;
}

// blocks b, b + 1 (b even) of the factor in one launch, and the rows below them in one launch (see the kernels)
static int gs_diag256(gsum_ctx* ctx, hipStream_t s, gsum_mat* m, int b) {
    gs_slot* sl = ctx->cur;
    const int64_t c = (int64_t)b * GS_NB;
    unsigned long long* stamps = ctx->diag_stamps ? ctx->dstamps : (unsigned long long*)nullptr;
    const int rec = gs_prof_begin(ctx, s, GS_PROF_DIAG, 8.0 * GS_NB * GS_NB * GS_NB / 3.0);
    hipLaunchKernelGGL(k_potrf_diag256, dim3(1), dim3(256), 0, s, m->A + c * m->ld + c, m->ld, m->Ltab + (size_t)b * GS_LTAB,
                       m->Lsib + (size_t)(b / 2) * GS_LSIB, m->logdet + b, sl->dinfo, (int)c, m->diag0 + c, stamps);
    gs_prof_end(ctx, s, rec);
    GS_CHECK(hipGetLastError());
    return 0;
}

static int gs_panel256(gsum_ctx* ctx, hipStream_t s, const gsum_mat* m, int b, double* P, int64_t ldp, int64_t M) {
    if (M <= 0) return 0;
    const int rec = gs_prof_begin(ctx, s, GS_PROF_PANEL, 4.0 * (double)M * GS_NB * GS_NB);
    hipLaunchKernelGGL(k_panel256, dim3((unsigned)((M + 15) / 16)), dim3(64), 0, s, P, ldp, (int)M, m->Ltab + (size_t)b * GS_LTAB,
                       m->Lsib + (size_t)(b / 2) * GS_LSIB, m->Ltab + (size_t)(b + 1) * GS_LTAB, ctx->kst_ptr, ctx->panel_stats);
    ctx->kst_ptr = nullptr;
    gs_prof_end(ctx, s, rec);
    GS_CHECK(hipGetLastError());
    return 0;
}

// the explicit 128 x 128 inverses of the diagonal blocks, for the consumers that want them (cho_solve's back-substitution)
static int gs_need_linv(gsum_ctx* ctx, hipStream_t s, gsum_mat* m) {
    if (m->have_linv) return 0;
    hipLaunchKernelGGL(k_trtri_blocks, dim3((unsigned)m->T), dim3(256), 0, s, m->Ltab, m->Linv);
    GS_CHECK(hipGetLastError());
    m->have_linv = true;
    return 0;
}

// ---- persistent-chain schedule (see k_chain) ------------------------------------------------------------------------
// Do kernels of two streams of this process run side by side?  The chain kernel waits for flags that host-enqueued kernels
// on other streams set, and they wait for its flags: under a tool that serialises dispatches (rocprofv3's kernel trace does)
// that would stall until the in-kernel timeout.  One spinning wave on one stream, the word it waits for written from another;
// 20 ms at most, once per context.
// The probe is the schedule's own triangle: a kernel that spins on the CHAIN's stream (sp) while the main stream (sm) and the
// auxiliary stream (sa) each deliver a word to it.  (Round 3 probed sm against sa only; with more high-priority streams in the
// process than hardware queues of that priority -- the groups' chain streams of a batch call created first -- sp and sa came to share
// a queue, the probe passed and the first single factorisation timed out: found by bench.py's own single-evaluation leg.)
static int gs_chain_probe(gsum_ctx* ctx, gs_slot* sl) {
    if (ctx->chain_probe != 0) return 0;
    unsigned* d = (unsigned*)ctx->dstamps + 64;                 // words 64.. of the 64 x u64 stamp buffer: unused by the stamps' 8 x u64
    GS_CHECK(hipMemsetAsync(d, 0, 4 * sizeof(unsigned), sl->sm));
    GS_CHECK(hipEventRecord(sl->evFork, sl->sm));
    GS_CHECK(hipStreamWaitEvent(sl->sp, sl->evFork, 0));
    GS_CHECK(hipStreamWaitEvent(sl->sa, sl->evFork, 0));
    hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(64), 0, sl->sp, (const unsigned*)d, (const unsigned*)(d + 1), 2000000ull, d + 2);     // <= 20 ms
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, sl->sm, d, 1u);
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, sl->sa, d + 1, 1u);
    GS_CHECK(hipGetLastError());
    GS_CHECK(hipStreamSynchronize(sl->sp));
    GS_CHECK(hipStreamSynchronize(sl->sa));
    GS_CHECK(hipStreamSynchronize(sl->sm));
    unsigned seen = 0;
    GS_CHECK(hipMemcpy(&seen, d + 2, sizeof(unsigned), hipMemcpyDeviceToHost));
    ctx->chain_probe = seen ? 1 : -1;
    return 0;
}

// the second high-priority stream of the persistent-chain schedule (rest of the panel, near update)
static int gs_aux_stream(gsum_ctx* ctx, gs_slot* sl) {
    if (!sl->sa) {
        GS_CHECK(hipStreamCreateWithPriority(&sl->sa, hipStreamNonBlocking, ctx->prio_hi));
        GS_CHECK(hipEventCreateWithFlags(&sl->evC, hipEventDisableTiming));
        GS_CHECK(hipEventCreateWithFlags(&sl->evS, hipEventDisableTiming));
    }
    return 0;
}

static int gs_chain_resources(gsum_ctx* ctx, gs_slot* sl, gsum_mat* m) {
    if (gs_aux_stream(ctx, sl)) return -1;
    const int S = m->T / 2;
    if (!m->cflags) GS_CHECK(hipMalloc((void**)&m->cflags, (size_t)(gs_fl_count(S) + S + 4) * sizeof(unsigned)));    // flags | fbwant[S]
    if (!m->cdump) GS_CHECK(hipMalloc((void**)&m->cdump, (size_t)2 * GS_CH_GMAX * 16 * 256 * sizeof(double)));
    if (!m->cstamps) {       // S x 16 chain stamps | S x 4 launch starts (preset to all ones: atomicMin) | S x 4 launch ends
        GS_CHECK(hipMalloc((void**)&m->cstamps, (size_t)S * (GS_CH_STAMPS + GS_CH_KSTAMPS) * sizeof(unsigned long long)));
        GS_CHECK(hipMemset(m->cstamps, 0, (size_t)S * (GS_CH_STAMPS + GS_CH_KSTAMPS) * sizeof(unsigned long long)));
    }
    if (!ctx->lds_attr_done.count((const void*)k_chain)) {
        GS_CHECK(hipFuncSetAttribute((const void*)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(GS_CH_LDS_DOUBLES * sizeof(double))));
        ctx->lds_attr_done.insert((const void*)k_chain);
    }
    return 0;
}

static bool gs_chain_wanted(const gsum_ctx* ctx, const gsum_mat* m) {
    if (ctx->chain_persist == 0 || ctx->chain_events_needed) return false;
    if (m->T < 4 || (m->T & 1)) return false;
    return ctx->chain_persist > 0 || m->np >= ctx->chain_min_np;
}

// Outer step s (panel columns [c0, c0 + 256), trailing matrix from r2 = c0 + 256), K = 256 everywhere:
//   k_chain            diagonal super-block, the window rows [r2, r2 + W) of the panel, C[window rows][r2, r2 + 256)
//   sa: rest(s)        rows >= r2 + W of the panel (k_panel256, gated on T1[s])                       -> evP[s]
//       A(s)           C[rows >= r2 + W][r2, r2 + 256)            gated on the window's first 16 row groups and on FB[s - 1]  -> FA[s]
//   sm: B(s) + Far(s)  lower tiles of C[rows, columns >= r2 + 256] in one launch, gated on the whole window, after rest(s); the tiles
//                      of its first 256 columns (B) first, counted in FB[s]
static int gs_potrf_chain(gsum_ctx* ctx, gsum_mat* m) {
    gs_slot* sl = ctx->cur;
    const int T = m->T, S = T / 2;
    const int W = ctx->chain_rows >= 512 ? 512 : 256;
    const int64_t ld = m->ld, naug = m->np + GS_BORDER;
    double* A = m->A;
    unsigned* fl = m->cflags;
    hipStream_t sm = sl->sm, sp = sl->sp, sa = sl->sa;
    GS_CHECK(hipMemsetAsync(fl, 0, (size_t)gs_fl_count(S) * sizeof(unsigned), sm));
    // The per-step table fbwant[] (how many first-column tiles of step s's trailing update the chain waits for) is part of what the
    // chain kernel reads: it is computed and uploaded HERE, in stream order ahead of the launch (round 3 uploaded it after the launch
    // with a synchronous copy on the null stream -- nothing ordered the two, and a delayed host could have let the chain read a stale
    // table: ADVICE round 3).  The host copy lives in the matrix object: it outlives the asynchronous copy.
    struct Plan { int kind; unsigned fb; };          // kind 0: banded B + Far, 1: near-512 (even, lazy), 2: B then Far K = 512 (odd, lazy)
    std::vector<Plan> plan((size_t)S, Plan{0, 0u});
    const bool lazy = ctx->chain_lazy > 0 || (ctx->chain_lazy < 0 && m->np >= 10240);
    const bool near256 = lazy && ctx->chain_lazy != 1;        // 2 / auto: only the next-but-one panel's 256 columns are "near" (the batch schedule's lazy_far = 2)
    const int NB = 1;                 // (row bands of the trailing update on streams of their own were measured in round 3 and removed in round 4:
                                      //  2 / 3 / 4 bands 5.84 / 5.87 / 6.79 ms against 5.45 with one at n = 8192)
    int64_t bound[6];
    bound[0] = 0;
    for (int p = 1; p < NB; ++p) bound[p] = (int64_t)(std::sqrt((double)p / NB) * (double)m->np / 256.0 + 0.5) * 256;
    bound[NB] = naug;
    auto first_tiles_of = [&](int s) {               // tiles of the first 256 trailing columns over all bands of step s
        const int64_t r3 = 256 * (int64_t)(s + 2);
        unsigned cnt = 0;
        for (int p = 0; p < NB; ++p) {
            const int64_t lo = std::max(bound[p], r3), hi = bound[p + 1];
            if (lo >= hi) continue;
            if (lo > r3) cnt += 4u * (unsigned)((hi - lo + 127) / 128);               // rectangle: all its first four column tiles
            else cnt += 4u * (unsigned)((hi - lo + 127) / 128) - 2u;                   // the triangle that starts at r3
        }
        return cnt;
    };
    {
        bool deferred = false;
        for (int s = 0; s + 1 < S; ++s) {
            const int64_t r3 = 256 * (int64_t)(s + 2), m3 = naug - r3;
            if (m3 <= 0) continue;
            const unsigned tm = (unsigned)((m3 + 127) / 128);
            if (deferred) {
                plan[s] = Plan{2, near256 ? 4u * tm - 2u : 4u * tm};
                deferred = false;
            } else if (lazy && m3 >= 1024 + GS_BORDER && s + 2 < S) {
                plan[s] = Plan{1, 4u * tm};
                deferred = true;
            } else {
                plan[s] = Plan{0, first_tiles_of(s)};
            }
        }
    }
    unsigned* fbw = fl + gs_fl_count(S);
    const int fb_key = (W * 2 + (lazy ? 1 : 0)) * 8 + NB + (near256 ? 1024 : 0);
    if (m->fbwant_key != fb_key) {
        GS_CHECK(hipStreamSynchronize(sm));                      // (a previous upload from the same host buffer has completed)
        m->fbwant_host.resize((size_t)S);
        for (int s = 0; s < S; ++s) m->fbwant_host[s] = plan[s].fb;
        GS_CHECK(hipMemcpyAsync(fbw, m->fbwant_host.data(), m->fbwant_host.size() * sizeof(unsigned), hipMemcpyHostToDevice, sm));
        m->fbwant_key = fb_key;
    }
    GS_CHECK(hipEventRecord(sl->evFork, sm));
    GS_CHECK(hipStreamWaitEvent(sp, sl->evFork, 0));
    GS_CHECK(hipStreamWaitEvent(sa, sl->evFork, 0));
    gs_chain_args ca;
    ca.A = A; ca.ld = ld; ca.np = (int)m->np; ca.naug = (int)naug; ca.S = S; ca.W = W;
    ca.Ltab = m->Ltab; ca.Lsib = m->Lsib; ca.logdet = m->logdet; ca.diag0 = m->diag0; ca.info = sl->dinfo;
    ca.dump = m->cdump; ca.flags = fl; ca.fbwant = fl + gs_fl_count(S); ca.stamps = ctx->chain_stamps ? m->cstamps : nullptr;
    ca.test_abort = ctx->chain_test_abort;
    ctx->chain_test_abort = 0;
    {
        const int rec = gs_prof_begin(ctx, sp, GS_PROF_DIAG, (double)T * GS_NB * GS_NB * GS_NB / 3.0);
        hipLaunchKernelGGL(k_chain, dim3((unsigned)(1 + W / 64)), dim3(256), GS_CH_LDS_DOUBLES * sizeof(double), sp, ca);
        gs_prof_end(ctx, sp, rec);
        GS_CHECK(hipGetLastError());
    }
    // nothing of the other streams is dispatched before every workgroup of the chain is resident (see k_wait_flag); the main
    // stream follows sa through evP[0]
    hipLaunchKernelGGL(k_wait_flag, dim3(1), dim3(64), 0, sa, (const unsigned*)(fl + GS_FL_RESIDENT), (unsigned)(1 + W / 64),
                       (const unsigned*)nullptr, 0u, fl);
    GS_CHECK(hipGetLastError());
    // the stream waits for chain flags (one spinning wave; see k_wait_flag)
    auto wait1 = [&](hipStream_t st, int kind, int s, unsigned want) {
        hipLaunchKernelGGL(k_wait_flag, dim3(1), dim3(64), 0, st, (const unsigned*)(fl + gs_fl(kind, S, s)), want, (const unsigned*)nullptr, 0u, fl);
    };
    auto wait2 = [&](hipStream_t st, int kind, int s, unsigned want, int kind2, int s2, unsigned want2) {
        hipLaunchKernelGGL(k_wait_flag, dim3(1), dim3(64), 0, st, (const unsigned*)(fl + gs_fl(kind, S, s)), want,
                           (const unsigned*)(fl + gs_fl(kind2, S, s2)), want2, fl);
    };
    unsigned long long* kst0 = nullptr;          // launch stamps (diagnostics): a (first start, last end) pair per launch, four per step
    if (ctx->chain_stamps) {
        // a launch writes kst[0] (atomicMin) and kst[1] (atomicMax): interleave (start, end) pairs, starts preset to all ones
        kst0 = m->cstamps + (size_t)S * GS_CH_STAMPS;
        std::vector<unsigned long long> init((size_t)S * GS_CH_KSTAMPS);
        for (size_t i = 0; i < init.size(); ++i) init[i] = (i & 1) ? 0ull : ~0ull;
        GS_CHECK(hipMemcpyAsync(kst0, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, sm));
        GS_CHECK(hipStreamSynchronize(sm));      // (diagnostic mode only: the host vector goes out of scope)
    }
    auto kstamp = [&](int s, int which) { ctx->kst_ptr = kst0 ? kst0 + ((size_t)s * 4 + which) * 2 : nullptr; };
    // B and Far are ONE launch on the main stream (k_gemm_ld3, nfirst): B's tiles take the first block ids, start on an empty chip
    // the moment the previous trailing update ends, are stored write-through and counted in FB[s]; Far's tiles follow in the same
    // grid.  sa keeps rest -> A, ~70 us per step.  (Measured on the way: rest / A / B on sa with B waiting for Far(s - 1) by event:
    // the cycle rest -> A -> B -> rest, ~135 us, bound steps 11-20, 5.61 ms; the panel on a stream of its own with two more events
    // per step made every cross-stream wait 60-90 us, 6.9 ms; B as its own launch in front of Far on the main stream idled the chip
    // for ~45 us per step in the first third, 5.83 ms.)
    // LAZY FAR UPDATES (chain_lazy; the batch schedule's idea, worth far more here): one factorisation alone runs its trailing
    // updates exclusively, and an exclusive K = 256 launch spends 15-20 % of its time on C reads and stores nothing else hides
    // (47.5 TF/s at M = 7936 against 55 at K = 512: profiles/r03_bulk_cphase.log).  So even steps update only the 512 columns the
    // next two panels live in ("near", K = 256, rectangular) and the following odd step applies both panels to everything right of
    // them in ONE K = 512 pass after its own B columns.  Per element the same products in the same order: bit-identical.
    // ROW BANDS (chain_bands): the trailing update B + Far of a step is cut into row bands with boundaries fixed in absolute
    // coordinates (equal areas of the whole triangle: X_p = n sqrt(p / NB), rounded to 256), band p on stream p.  A tile of the
    // trailing matrix depends on its own previous version and on the panel only, so band p of step s + 1 may start when band p
    // of step s is done, whatever the other bands do: the launches of one step no longer end at a chip-wide barrier, and one
    // band's tail overlaps another's bulk -- what sixteen evaluations in flight do for a batch (52 TF/s of Cholesky flops there
    // against 42-44 for one factorisation's exclusive launches).  A band = a rectangle (columns left of its own rows) + a triangle.
    hipStream_t sbd[1] = {sm};
    for (int s = 0; s + 1 < S; ++s) {             // the last outer step has nothing below its window: the chain does all of it
        const int k = 2 * s;
        const int64_t c0 = 256 * (int64_t)s, r2 = c0 + 256, wend = std::min<int64_t>(r2 + W, naug), mr = naug - wend;
        const int Gs = (int)((wend - r2) / 16);
        if (mr > 0) {
            wait1(sa, GS_FL_T1, s, 1u);
            kstamp(s, 0);
            if (gs_panel256(ctx, sa, m, k, A + wend * ld + c0, ld, mr)) return -1;
        }
        // the whole window solved (operands of the trailing update, and of A): the panel of step s is complete ...
        wait1(sa, GS_FL_WALL, s, (unsigned)Gs);
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, sa, fl + gs_fl(GS_FL_RP, S, s), 1u);
        // ... and, for A, B(s - 1) (the same region of C)
        if (mr > 0) {
            if (s > 0) wait1(sa, GS_FL_FB, s - 1, plan[s - 1].fb);
            kstamp(s, 1);
            if (gs_gemm(ctx, sa, GS_BULK, A + wend * ld + r2, ld, A + wend * ld + c0, ld, A + r2 * ld + c0, ld, mr, 256, 256, 0, 1, -1.0)) return -1;
        }
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, sa, fl + gs_fl(GS_FL_FA, S, s), 1u);
        GS_CHECK(hipGetLastError());
        const int64_t r3 = r2 + 256, m3 = naug - r3;
        if (m3 <= 0) continue;
        unsigned* fbp = fl + gs_fl(GS_FL_FB, S, s);
        double* P3 = A + r3 * ld + c0;               // panel rows r3.., this step's 256 columns
        if (plan[s].kind == 0) {
            for (int p = 0; p < NB; ++p) {
                const int64_t lo = std::max(bound[p], r3), hi = bound[p + 1];
                if (lo >= hi) continue;
                hipStream_t sb = sbd[p];
                wait1(sb, GS_FL_RP, s, 1u);
                double* Plo = A + lo * ld + c0;          // panel rows of this band
                if (lo > r3) {
                    // rectangle: rows [lo, hi) x columns [r3, lo); its first 256 columns are B's
                    ctx->first_tiles = (int)(4 * ((hi - lo + 127) / 128));
                    ctx->first_done = fbp;
                    if (p == NB - 1) kstamp(s, 3);
                    if (gs_gemm(ctx, sb, GS_BULK, A + lo * ld + r3, ld, Plo, ld, P3, ld, hi - lo, lo - r3, 256, 0, 1, -1.0)) return -1;
                    if (gs_gemm(ctx, sb, GS_BULK, A + lo * ld + lo, ld, Plo, ld, Plo, ld, hi - lo, hi - lo, 256, 1, 1, -1.0)) return -1;
                } else {
                    // the band the trailing matrix starts in: a triangle from r3, first-256-column tiles first
                    ctx->first_tiles = (int)(4 * ((hi - lo + 127) / 128) - 2);
                    ctx->first_done = fbp;
                    if (p == NB - 1) kstamp(s, 3); else kstamp(s, 2);
                    if (gs_gemm(ctx, sb, GS_BULK, A + lo * ld + lo, ld, Plo, ld, Plo, ld, hi - lo, hi - lo, 256, 1, 1, -1.0)) return -1;
                }
            }
            continue;
        }
        wait1(sm, GS_FL_RP, s, 1u);
        ctx->first_tiles = (int)plan[s].fb;
        ctx->first_done = fbp;
        if (plan[s].kind == 1) {
            // near region only: rows >= r3, columns [r3, r3 + 512) -- or just [r3, r3 + 256); algorithmic work = the lower trapezoid
            const double wn = near256 ? 256.0 : 512.0;
            ctx->next_algo_flops = 256.0 * (2.0 * (double)m3 * wn - wn * (wn - 1.0));
            kstamp(s, 3);
            if (gs_gemm(ctx, sm, GS_BULK, A + r3 * ld + r3, ld, P3, ld, P3, ld, m3, near256 ? 256 : 512, 256, 0, 1, -1.0)) return -1;
        } else if (near256) {
            // everything from column r3 on takes the previous panel and this one together (512 contiguous panel columns), the tiles of its first 256 columns first
            double* P4 = A + r3 * ld + (c0 - 256);
            kstamp(s, 3);
            if (gs_gemm(ctx, sm, GS_BULK, A + r3 * ld + r3, ld, P4, ld, P4, ld, m3, m3, 512, 1, 1, -1.0)) return -1;
        } else {
            // columns [r3, r3 + 256): this panel only (they had the previous one as "near") ...
            ctx->next_algo_flops = 256.0 * (2.0 * (double)m3 * 256.0 - 256.0 * 255.0);
            kstamp(s, 2);
            if (gs_gemm(ctx, sm, GS_BULK, A + r3 * ld + r3, ld, P3, ld, P3, ld, m3, std::min<int64_t>(256, m3), 256, 0, 1, -1.0)) return -1;
            // ... everything right of them: the previous panel and this one together (512 contiguous panel columns)
            const int64_t r4 = r3 + 256, m4 = naug - r4;
            if (m4 > 0) {
                double* P4 = A + r4 * ld + (c0 - 256);
                kstamp(s, 3);
                if (gs_gemm(ctx, sm, GS_BULK, A + r4 * ld + r4, ld, P4, ld, P4, ld, m4, m4, 512, 1, 1, -1.0)) return -1;
            }
        }
    }
    GS_CHECK(hipEventRecord(sl->evC, sp));
    GS_CHECK(hipEventRecord(sl->evS, sa));
    GS_CHECK(hipStreamWaitEvent(sm, sl->evC, 0));
    GS_CHECK(hipStreamWaitEvent(sm, sl->evS, 0));
    // a chain that gave up (flags[0] == 1) reports through the info word: INT_MAX is no LAPACK index
    hipLaunchKernelGGL(k_chain_status, dim3(1), dim3(64), 0, sm, (const unsigned*)fl, sl->dinfo);
    GS_CHECK(hipGetLastError());
    m->factored = true;
    return 0;
}

static int gs_potrf(gsum_ctx* ctx, gsum_mat* m) {
    const int T = m->T;
    gs_slot* sl = ctx->cur;
    if (gs_potrf_events(ctx, sl, T)) return -1;
    m->have_linv = false;                      // (explicit block inverses: built on demand, gs_need_linv)
    m->have_ltab = true;
    m->solved_k = -1;
    const int64_t ld = m->ld, naug = m->np + GS_BORDER;
    double* A = m->A;
    GS_CHECK(hipMemsetAsync(sl->dinfo, 0, sizeof(int), sl->sm));
    {
        const int rec = gs_prof_begin(ctx, sl->sm, GS_PROF_OTHER, 0.0);
        hipLaunchKernelGGL(k_save_diag, dim3((unsigned)((m->np + 255) / 256)), dim3(256), 0, sl->sm, A, ld, (int)m->np, m->diag0);
        gs_prof_end(ctx, sl->sm, rec);
    }
    GS_CHECK(hipGetLastError());
    // look-ahead shortens ONE factorisation; with several in flight the others already fill the GPU and the
    // extra look-ahead launches only cost (measured: 3 in flight without look-ahead beats 4 with)
    // Only slot 0 ever runs a look-ahead schedule (the gradient batch's other slots run everything on their one stream).
    const bool la = ctx->lookahead != 0 && ctx->batch_active < 3 && sl == &ctx->slots[0];
    ctx->bulk_pad_now = false;
    if (la && gs_panel_stream(ctx, sl)) return -1;
    if (la && gs_chain_wanted(ctx, m)) {
        if (gs_chain_resources(ctx, sl, m)) return -1;
        if (gs_chain_probe(ctx, sl)) return -1;
        if (ctx->chain_probe > 0) return gs_potrf_chain(ctx, m);
    }
    hipStream_t sp = la ? sl->sp : sl->sm;
    hipStream_t sm = sl->sm, sb = sl->sm;
    if (la) {
        GS_CHECK(hipEventRecord(sl->evFork, sm));
        GS_CHECK(hipStreamWaitEvent(sp, sl->evFork, 0));
    }
    // panel GEMMs (TRSM against the block inverse, sibling column) stay on the low-latency 32x128 tile in every mode.
    // (In a batch the LDS-direct 128x128 tile is 1 % cheaper overall, but then one kernel symbol would serve two
    // roles and rocprofv3's per-kernel average would no longer be the bulk update's.)
    const int ccfg = 1;
    unsigned long long* stamps = ctx->diag_stamps ? ctx->dstamps : (unsigned long long*)nullptr;
    int prev = -1;                                   // outer step whose bulk update is still in flight
    bool deferred = false;                           // batch mode: the far region still owes the previous panel's update
    for (int k = 0; k < T; k += 2) {
        const bool two = k + 1 < T;
        const int64_t c0 = (int64_t)k * GS_NB, c1 = c0 + GS_NB;
        const int64_t r2 = two ? c1 + GS_NB : c1;   // first row / column of the trailing matrix
        const int Kp = two ? 2 * GS_NB : GS_NB;
        double* Pa = A + c1 * ld + c0;              // rows below diagonal block k, border included
        if (two && (ctx->chain_fused > 0 || (ctx->chain_fused < 0 && ctx->batch_active >= 3))) {
            // both diagonal blocks in one launch, then both panels of the rows below in one
            if (gs_diag256(ctx, sp, m, k)) return -1;
            if (gs_panel256(ctx, sp, m, k, A + r2 * ld + c0, ld, naug - r2)) return -1;
        } else {
            // ---- sub-step a
            if (gs_diag(ctx, sp, m, k)) return -1;
            if (gs_trsm_rows(ctx, sp, m, k, Pa, ld, naug - c1)) return -1;
            if (two) {
                // block column k+1 (rows c1..) -= P_a P_a[first 128 rows]^T, then its own diag + trsm
                if (gs_gemm(ctx, sp, ccfg, A + c1 * ld + c1, ld, Pa, ld, Pa, ld, naug - c1, GS_NB, GS_NB, 0, 1, -1.0)) return -1;
                if (gs_diag(ctx, sp, m, k + 1)) return -1;
                double* Pb = A + r2 * ld + c1;
                if (gs_trsm_rows(ctx, sp, m, k + 1, Pb, ld, naug - r2)) return -1;
            }
        }
        // ---- trailing update with the whole panel: rows r2.., columns c0..c0+Kp-1
        double* P = A + r2 * ld + c0;
        const int64_t mrest = naug - r2;            // >= 16 (the border)
        if (!la) {
            // Batch mode (latency is irrelevant, the bulk kernel's fixed per-launch cost is not): lazy far updates.
            // Even outer steps update only the 512 columns the next two panels live in (K = 256) and defer the rest;
            // the following odd step applies both panels to the deferred region in ONE pass (K = 512: half the C
            // traffic and launch overhead there).  Every element still subtracts the same products in the same
            // ascending-k order, so results do not change.
            const int64_t w2 = 2 * GS_NB;
            // measured (lazy_far = 1 against none): -2.3 % per evaluation at n = 8192, neutral at 7000, +8 % (extra launches) at 4096 and below;
            // lazy_far = 2 against 1 at n = 8192, same process: 303.4-303.8 against 297.6-298.7 evals/s (+1.9 %), profiles/r03_lazy_far2_ab.log
            const bool full_next = ctx->lazy_far && m->np >= ctx->lazy_min_np && two && r2 + 2 * w2 <= m->np;   // a full panel follows, and one more
            if (!deferred && full_next) {
                // near region only: rows >= r2, columns [r2, r2 + 512) -- or, lazy_far = 2, just the next panel's 256 columns: the panel after that then
                // takes both updates in the K = 512 launch below, which moves two thirds of the near region's flops out of skinny K = 256 launches and
                // saves one launch per pair of steps; algorithmic work = the lower trapezoid
                const int64_t wn = ctx->lazy_far == 2 ? w2 : 2 * w2;
                ctx->next_algo_flops = (double)Kp * (2.0 * (double)mrest * wn - (double)wn * (wn - 1));
                if (gs_gemm(ctx, sm, GS_BULK, A + r2 * ld + r2, ld, P, ld, P, ld, mrest, wn, Kp, 0, 1, -1.0)) return -1;
                deferred = true;
                continue;
            }
            if (deferred && ctx->lazy_far == 2) {
                // everything from column r2 on: the previous panel and this one together (contiguous 512 columns), lower triangle
                double* P2 = A + r2 * ld + (c0 - w2);
                if (gs_gemm(ctx, sm, GS_BULK, A + r2 * ld + r2, ld, P2, ld, P2, ld, mrest, mrest, (int)(w2 + Kp), 1, 1, -1.0)) return -1;
                deferred = false;
                continue;
            }
            if (deferred) {
                // columns [r2, r2 + 256): this panel only (they had the previous one as "near")
                ctx->next_algo_flops = (double)Kp * (2.0 * (double)mrest * w2 - (double)w2 * (w2 - 1));
                if (gs_gemm(ctx, sm, GS_BULK, A + r2 * ld + r2, ld, P, ld, P, ld, mrest, w2, Kp, 0, 1, -1.0)) return -1;
                // everything right of them: the previous panel and this one together (contiguous 512 columns)
                const int64_t rf = r2 + w2, mf = naug - rf;
                double* P2 = A + rf * ld + (c0 - w2);
                if (gs_gemm(ctx, sm, GS_BULK, A + rf * ld + rf, ld, P2, ld, P2, ld, mf, mf, (int)(w2 + Kp), 1, 1, -1.0)) return -1;
                deferred = false;
                continue;
            }
            if (gs_gemm(ctx, sm, GS_BULK, A + r2 * ld + r2, ld, P, ld, P, ld, mrest, mrest, Kp, 1, 1, -1.0)) return -1;
            continue;
        }
        GS_CHECK(hipEventRecord(sl->evP[k], sp));
        if (r2 < m->np) {
            const int64_t wn = std::min<int64_t>(2 * GS_NB, m->np - r2);     // width of the next panel
            // look-ahead columns: need the previous bulk update to have finished with THEM (evA: see below)
            if (prev >= 0) GS_CHECK(hipStreamWaitEvent(sp, sl->evA[prev], 0));
            if (gs_gemm(ctx, sp, ccfg, A + r2 * ld + r2, ld, P, ld, P, ld, mrest, wn, Kp, 0, 1, -1.0)) return -1;
            const int64_t r3 = r2 + wn, m3 = naug - r3;
            double* P3 = A + r3 * ld + c0;
            GS_CHECK(hipStreamWaitEvent(sb, sl->evP[k], 0));
            // The bulk update goes out in two launches: first the 256 columns the panel AFTER the next one lives in, then
            // everything right of them.  The next step's look-ahead update waits for the first only, so the chain is a
            // whole outer step ahead of the bulk stream instead of starting when the previous bulk update ends: in the
            // first third of a factorisation (bulk-bound) the chain then hides under the bulk update completely.
            const int64_t wa = std::min<int64_t>(2 * GS_NB, m->np - r3);
            if (ctx->la_depth2 && wa > 0 && m3 > wa) {
                ctx->next_algo_flops = (double)Kp * (2.0 * (double)m3 * wa - (double)wa * (wa - 1));
                if (gs_bulk_la(ctx, sb, GS_BULK, A + r3 * ld + r3, ld, P3, ld, P3, ld, m3, wa, Kp, 0, 1, -1.0)) return -1;
                GS_CHECK(hipEventRecord(sl->evA[k], sb));
                const int64_t r4 = r3 + wa, m4 = naug - r4;
                double* P4 = A + r4 * ld + c0;
                if (gs_bulk_la(ctx, sb, GS_BULK, A + r4 * ld + r4, ld, P4, ld, P4, ld, m4, m4, Kp, 1, 1, -1.0)) return -1;
            } else {
                if (gs_bulk_la(ctx, sb, GS_BULK, A + r3 * ld + r3, ld, P3, ld, P3, ld, m3, m3, Kp, 1, 1, -1.0)) return -1;
                GS_CHECK(hipEventRecord(sl->evA[k], sb));
            }
            GS_CHECK(hipEventRecord(sl->evM[k], sb));
            prev = k;
        } else {
            // last panel: only the 16x16 corner (the Gram matrix) is left
            GS_CHECK(hipStreamWaitEvent(sm, sl->evP[k], 0));
            if (gs_gemm(ctx, sm, GS_BULK, A + r2 * ld + r2, ld, P, ld, P, ld, mrest, mrest, Kp, 1, 1, -1.0)) return -1;
        }
    }
    m->factored = true;
    return 0;
}

static int gs_finalize(gsum_ctx* ctx, gsum_mat* m) {
    const int rec = gs_prof_begin(ctx, ctx->cur->sm, GS_PROF_OTHER, 0.0);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, ctx->cur->sm, m->A, m->ld, (int)m->np, m->logdet, m->T, ctx->cur->dinfo,
                       ctx->cur->dres);
    gs_prof_end(ctx, ctx->cur->sm, rec);
    GS_CHECK(hipGetLastError());
    GS_CHECK(hipMemcpyAsync(ctx->cur->hres, ctx->cur->dres, 258 * sizeof(double), hipMemcpyDeviceToHost, ctx->cur->sm));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

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
    auto fail = [&](const char* what, hipError_t err) {
        g_init_error = std::string(what) + ": " + hipGetErrorString(err);
        delete ctx;
        return -1;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
    (void)hipDeviceGetStreamPriorityRange(&ctx->prio_lo, &ctx->prio_hi);   // hi = numerically lowest
    if (gs_need_slots(ctx, 1)) {
        g_init_error = ctx->err;
        delete ctx;
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
        delete ctx;
        return -1;
    }
    if ((e = hipStreamCreateWithPriority(&ctx->wave.g[2].sc, hipStreamNonBlocking, ctx->prio_hi)) != hipSuccess) return fail("hipStreamCreateWithPriority", e);
    ctx->wave.g[2].own_sc = true;
    ctx->wave.sb = ctx->cur->sm;
    if ((e = hipMalloc((void**)&ctx->dstamps, 64 * sizeof(unsigned long long))) != hipSuccess) return fail("hipMalloc", e);
    (void)hipMemset(ctx->dstamps, 0, 64 * sizeof(unsigned long long));
    const char* la = getenv("GSUM_LOOKAHEAD");
    if (la) ctx->lookahead = atoi(la);
    const char* pg = getenv("GSUM_PIVOT_GUARD_ULPS");
    if (pg) {
        const double g = (double)std::max(0, std::min(1024, atoi(pg))) * 2.220446049250313e-16;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(gs_pivot_guard), &g, sizeof g);
    }
    const char* cp = getenv("GSUM_CHAIN_PERSIST");
    if (cp) ctx->chain_persist = atoi(cp) < 0 ? -1 : (atoi(cp) != 0);
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
    else if (!strcmp(name, "chain_test_abort")) ctx->chain_test_abort = (int)std::max<int64_t>(0, value);
    else if (!strcmp(name, "chain_stamps")) ctx->chain_stamps = value != 0;
    else if (!strcmp(name, "batch_slots")) ctx->batch_slots = (int)std::max<int64_t>(1, std::min<int64_t>(GS_MAX_SLOTS, value));
    else if (!strcmp(name, "wave_shift")) ctx->wave_shift = (int)std::max<int64_t>(-1, value);
    else if (!strcmp(name, "wave_min")) ctx->wave_min = (int)std::max<int64_t>(1, value);
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
    else if (!strcmp(name, "wave_groups")) ctx->wave_groups = (int)std::max<int64_t>(1, std::min<int64_t>(GS_WV_GROUPS, value));
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

static int gs_upload_X(gsum_ctx* ctx, gs_inputs* I, const double* X, int64_t n, int d) {
    if (!X || n <= 0) GS_FAIL("X is NULL or empty");
    if (gs_reserve(ctx, &I->X, &I->X_cap, (size_t)n * d * sizeof(double))) return -1;
    GS_CHECK(hipMemcpyAsync(I->X, X, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice, ctx->cur->sm));
    I->n = n;
    I->d = d;
    return 0;
}

static int gs_upload_Z(gsum_ctx* ctx, gs_inputs* I, const double* Z, int64_t n, int k) {
    if (k < 0 || k > GSUM_MAX_RHS) GS_FAIL("k must be 0..GSUM_MAX_RHS");
    if (k > 0 && !Z) GS_FAIL("RHS is NULL");
    if (gs_reserve(ctx, &I->Z, &I->Z_cap, std::max<size_t>(8, (size_t)n * k * sizeof(double)))) return -1;
    if (k > 0) GS_CHECK(hipMemcpyAsync(I->Z, Z, (size_t)n * k * sizeof(double), hipMemcpyHostToDevice, ctx->cur->sm));
    I->k = k;
    return 0;
}

static int gs_check_series(gsum_ctx* ctx, const gsum_series_scale* sc);

// kernel(X[, Y]) -> host, optionally scaled like TruncationProcess.cov on the device before it leaves (sc != NULL)
static int gs_kernel_build_host(gsum_ctx* ctx, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d, const double* Y, int64_t m,
                                double diag_add, const gsum_series_scale* sc, const double* ref_x, const double* ratio_x, const double* ref_y,
                                const double* ratio_y, double* out) {
    if (!ctx) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (gs_check_desc(ctx, desc, d)) return -2;
    if (!X || !out || n <= 0) GS_FAIL("bad argument");
    const bool cross = Y != nullptr;
    const int64_t cols = cross ? m : n;
    if (cols <= 0) GS_FAIL("bad argument");
    if (sc && (gs_check_series(ctx, sc) || !ref_x || !ratio_x || (cross && (!ref_y || !ratio_y)))) {
        if (ctx->err.empty()) ctx->err = "series scaling needs ref / ratio for both point sets";
        return -2;
    }
    const int64_t ldo = (cols + 1) / 2 * 2;
    const size_t xb = (size_t)n * d * sizeof(double), yb = cross ? (size_t)m * d * sizeof(double) : 0;
    const size_t ob = (size_t)n * ldo * sizeof(double), vb = sc ? (size_t)2 * (n + cols) * sizeof(double) : 0;
    const size_t off_y = (xb + 255) / 256 * 256, off_o = off_y + (yb + 255) / 256 * 256, off_v = off_o + (ob + 255) / 256 * 256;
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, off_v + vb)) return -1;
    char* base = (char*)ctx->scratch;
    double* dXl = (double*)base;
    double* dYl = (double*)(base + off_y);
    double* dO = (double*)(base + off_o);
    hipStream_t s = ctx->cur->sm;
    GS_CHECK(hipMemcpyAsync(dXl, X, xb, hipMemcpyHostToDevice, s));
    if (cross) GS_CHECK(hipMemcpyAsync(dYl, Y, yb, hipMemcpyHostToDevice, s));
    if (cross ? gs_launch_build<true>(ctx, s, dO, ldo, dXl, dYl, n, m, n, ldo, d, desc, 0.0, 0)
              : gs_launch_build<false>(ctx, s, dO, ldo, dXl, nullptr, n, n, n, ldo, d, desc, diag_add, 0))
        return -1;
    if (sc) {
        double* v = (double*)(base + off_v);
        double *d_ref_r = v, *d_rat_r = v + n, *d_ref_c = v + 2 * n, *d_rat_c = v + 2 * n + cols;
        GS_CHECK(hipMemcpyAsync(d_ref_r, ref_x, (size_t)n * 8, hipMemcpyHostToDevice, s));
        GS_CHECK(hipMemcpyAsync(d_rat_r, ratio_x, (size_t)n * 8, hipMemcpyHostToDevice, s));
        GS_CHECK(hipMemcpyAsync(d_ref_c, cross ? ref_y : ref_x, (size_t)cols * 8, hipMemcpyHostToDevice, s));
        GS_CHECK(hipMemcpyAsync(d_rat_c, cross ? ratio_y : ratio_x, (size_t)cols * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_scale_series, dim3((unsigned)((cols + 255) / 256), (unsigned)n), dim3(256), 0, s, dO, ldo, (int)n, (int)cols,
                           d_ref_r, d_rat_r, d_ref_c, d_rat_c, *sc);
        GS_CHECK(hipGetLastError());
    }
    GS_CHECK(hipMemcpy2DAsync(out, (size_t)cols * sizeof(double), dO, (size_t)ldo * sizeof(double),
                              (size_t)cols * sizeof(double), (size_t)n, hipMemcpyDeviceToHost, s));
    GS_CHECK(hipStreamSynchronize(s));
    return 0;
}

int gsum_kernel_build(gsum_ctx* ctx, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d,
                      const double* Y, int64_t m, double diag_add, double* out) {
    return gs_kernel_build_host(ctx, desc, X, n, d, Y, m, diag_add, nullptr, nullptr, nullptr, nullptr, nullptr, out);
}

int gsum_kernel_build_series(gsum_ctx* ctx, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d, const double* Y, int64_t m,
                             double diag_add, const gsum_series_scale* sc, const double* ref_x, const double* ratio_x,
                             const double* ref_y, const double* ratio_y, double* out) {
    if (!ctx || !sc) return -2;
    return gs_kernel_build_host(ctx, desc, X, n, d, Y, m, diag_add, sc, ref_x, ratio_x, ref_y, ratio_y, out);
}

int gsum_kernel_build_dev(gsum_ctx* ctx, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d,
                          double diag_add, gsum_mat** out) {
    if (!ctx || !out) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (gs_check_desc(ctx, desc, d)) return -2;
    if (gs_upload_X(ctx, &ctx->op, X, n, d)) return -1;
    gsum_mat* m = nullptr;
    if (gs_mat_alloc(ctx, n, &m)) return -1;
    if (gs_build_into(ctx, ctx->cur->sm, m, desc, ctx->op.X, d, diag_add, ctx->build_lower_only) ||
        gs_set_border(ctx, ctx->cur->sm, m, nullptr, 0)) {
        gs_mat_release(m);
        return -1;
    }
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    *out = m;
    return 0;
}

int gsum_mat_from_host(gsum_ctx* ctx, const double* Ah, int64_t n, gsum_mat** out) {
    if (!ctx || !out || !Ah) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    gsum_mat* m = nullptr;
    if (gs_mat_alloc(ctx, n, &m)) return -1;
    hipError_t e = hipMemcpy2DAsync(m->A, (size_t)m->ld * sizeof(double), Ah, (size_t)n * sizeof(double),
                                    (size_t)n * sizeof(double), (size_t)n, hipMemcpyHostToDevice, ctx->cur->sm);
    if (e == hipSuccess && m->np > n) {
        hipLaunchKernelGGL(k_pad_identity, dim3((unsigned)((m->np + 255) / 256), (unsigned)(m->np - n)), dim3(256), 0,
                           ctx->cur->sm, m->A, m->ld, (int)n, (int)m->np);
        e = hipGetLastError();
    }
    if (e != hipSuccess || gs_set_border(ctx, ctx->cur->sm, m, nullptr, 0)) {
        gs_mat_release(m);
        if (e != hipSuccess) ctx->err = std::string("upload failed: ") + hipGetErrorString(e);
        return -1;
    }
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    *out = m;
    return 0;
}

int gsum_potrf_lower(gsum_ctx* ctx, gsum_mat* A, int64_t* info) {
    if (!ctx || !A || !info) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (A->factored) GS_FAIL("matrix is already factorised");
    if (gs_potrf(ctx, A)) return -1;
    if (gs_finalize(ctx, A)) return -1;
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    *info = (int64_t)ctx->cur->hres[257];
    if (*info == GS_INFO_CHAIN_ABORT) {
        ctx->chain_persist = 0;
        ++ctx->chain_aborts;
        A->factored = false;
        ctx->err = "the persistent chain schedule timed out (streams of this process do not run side by side); the matrix is "
                   "destroyed -- rebuild it and factorise again: the schedule is now switched off (option chain_persist = 0)";
        return GSUM_ERR_CHAIN_ABORT;        // a runtime failure the caller can recover from: rebuild the matrix, factorise again
    }
    if (*info > A->n) *info = A->n;     // cannot happen (identity padding), kept as a guard
    A->factored = (*info == 0);
    return 0;
}

// Forward substitution on the border rows against an existing factor (right-looking, block by block):
//   W_c = Z_c L_cc^-T ;  Z[:, rest] -= W_c L[rest, c]^T ;  corner accumulates -W W^T.
static int gs_border_solve(gsum_ctx* ctx, gsum_mat* m) {
    const int64_t ld = m->ld, naug = m->np + GS_BORDER;
    double* A = m->A;
    double* Brow = A + m->np * ld;
    for (int k = 0; k < m->T; ++k) {
        const int64_t c0 = (int64_t)k * GS_NB, r0 = c0 + GS_NB;
        if (gs_trsm_rows(ctx, ctx->cur->sm, m, k, Brow + c0, ld, GS_BORDER)) return -1;
        if (gs_gemm(ctx, ctx->cur->sm, 2, Brow + r0, ld, Brow + c0, ld, A + r0 * ld + c0, ld, GS_BORDER, naug - r0, GS_NB, 0, 1,
                    -1.0))
            return -1;
    }
    return 0;
}

// border rows <- (L^-1 RHS)^T, corner <- -W^T W; skipped when the rows already hold the solve of the same RHS (predict is
// called again and again with the same training residual: T x 2 dependent launches saved per call)
static int gs_border_prepare(gsum_ctx* ctx, gsum_mat* L, const double* RHS, int64_t n, int k) {
    const size_t cnt = (size_t)n * k;
    if (L->solved_k == k && L->solved_rhs.size() == cnt && !memcmp(L->solved_rhs.data(), RHS, cnt * sizeof(double))) return 0;
    L->solved_k = -1;
    if (gs_upload_Z(ctx, &ctx->op, RHS, n, k)) return -1;
    if (gs_set_border(ctx, ctx->cur->sm, L, ctx->op.Z, k)) return -1;
    if (gs_border_solve(ctx, L)) return -1;
    L->solved_rhs.assign(RHS, RHS + cnt);
    L->solved_k = k;
    return 0;
}

int gsum_forward_gram(gsum_ctx* ctx, gsum_mat* L, const double* RHS, int64_t n, int32_t k, double* G,
                      double* sum_log_diag) {
    if (!ctx || !L || !G || !sum_log_diag) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (!L->factored) GS_FAIL("forward_gram needs a factorised matrix");
    if (n != L->n) GS_FAIL("RHS has the wrong number of rows");
    if (k < 1 || k > GSUM_MAX_RHS) GS_FAIL("k must be 1..GSUM_MAX_RHS");
    GS_CHECK(hipMemsetAsync(ctx->cur->dinfo, 0, sizeof(int), ctx->cur->sm));
    if (gs_border_prepare(ctx, L, RHS, n, k)) return -1;
    if (gs_finalize(ctx, L)) return -1;
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) G[i * k + j] = ctx->cur->hres[i * 16 + j];
    *sum_log_diag = ctx->cur->hres[256];
    return 0;
}

int gsum_forward_solve(gsum_ctx* ctx, gsum_mat* L, const double* RHS, int64_t n, int32_t k, double* W) {
    if (!ctx || !L || !W) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (!L->factored) GS_FAIL("forward_solve needs a factorised matrix");
    if (n != L->n) GS_FAIL("RHS has the wrong number of rows");
    if (k < 1 || k > GSUM_MAX_RHS) GS_FAIL("k must be 1..GSUM_MAX_RHS");
    if (gs_border_prepare(ctx, L, RHS, n, k)) return -1;
    std::vector<double> rows((size_t)k * n);
    GS_CHECK(hipMemcpy2DAsync(rows.data(), (size_t)n * sizeof(double), L->A + L->np * L->ld, (size_t)L->ld * sizeof(double),
                              (size_t)n * sizeof(double), (size_t)k, hipMemcpyDeviceToHost, ctx->cur->sm));
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < k; ++c) W[i * k + c] = rows[(size_t)c * n + i];
    return 0;
}

// scipy.linalg.cho_solve((L, True), B) = L^-T (L^-1 B): the forward half is gs_border_solve (border rows = W^T), the
// backward half runs right-looking from the last block column to the first, in place on the border rows:
//   X_c^T = W_c^T L_cc^-1 ;  W^T[:, cols < c0] -= X_c^T L[c rows, cols < c0]        (k_back_first / k_back_step)
int gsum_cho_solve(gsum_ctx* ctx, gsum_mat* L, const double* B, int64_t n, int32_t k, double* X) {
    if (!ctx || !L || !B || !X) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (!L->factored) GS_FAIL("cho_solve needs a factorised matrix");
    if (n != L->n) GS_FAIL("B has the wrong number of rows");
    if (k < 1 || k > GSUM_MAX_RHS) GS_FAIL("k must be 1..GSUM_MAX_RHS");
    hipStream_t s = ctx->cur->sm;
    if (gs_border_prepare(ctx, L, B, n, k)) return -1;
    L->solved_k = -1;                  // the back-substitution below overwrites the border rows in place
    if (gs_need_linv(ctx, s, L)) return -1;
    double* Brow = L->A + L->np * L->ld;
    const int T = L->T;
    hipLaunchKernelGGL(k_back_first, dim3(1), dim3(256), 0, s, Brow, L->ld, L->Linv + (size_t)(T - 1) * GS_NB * GS_NB, (T - 1) * GS_NB);
    GS_CHECK(hipGetLastError());
    for (int c = T - 1; c >= 1; --c) {
        hipLaunchKernelGGL(k_back_step, dim3((unsigned)c), dim3(256), 0, s, L->A, L->ld, Brow, L->Linv, c);
        GS_CHECK(hipGetLastError());
    }
    std::vector<double> rows((size_t)k * n);
    GS_CHECK(hipMemcpy2DAsync(rows.data(), (size_t)n * sizeof(double), Brow, (size_t)L->ld * sizeof(double),
                              (size_t)n * sizeof(double), (size_t)k, hipMemcpyDeviceToHost, s));
    GS_CHECK(hipStreamSynchronize(s));
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < k; ++c) X[i * k + c] = rows[(size_t)c * n + i];
    return 0;
}

int gsum_tri_multiply(gsum_ctx* ctx, gsum_mat* L, const double* Z, int64_t n, int32_t k, double* out) {
    if (!ctx || !L || !Z || !out) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (!L->factored) GS_FAIL("tri_multiply needs a factorised matrix");
    if (n != L->n) GS_FAIL("Z has the wrong number of rows");
    if (k < 1 || k > GSUM_MAX_RHS) GS_FAIL("k must be 1..GSUM_MAX_RHS");
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, (size_t)2 * n * 16 * 8)) return -1;
    double* dZ16 = ctx->scratch;
    double* dOut = dZ16 + (size_t)n * 16;
    hipStream_t s = ctx->cur->sm;
    std::vector<double> pad((size_t)n * 16, 0.0);
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < k; ++c) pad[(size_t)i * 16 + c] = Z[i * k + c];
    GS_CHECK(hipMemcpyAsync(dZ16, pad.data(), pad.size() * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_tri_multiply, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, L->A, L->ld, (int)n, dZ16, dOut);
    GS_CHECK(hipGetLastError());
    GS_CHECK(hipMemcpyAsync(pad.data(), dOut, pad.size() * 8, hipMemcpyDeviceToHost, s));
    GS_CHECK(hipStreamSynchronize(s));
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < k; ++c) out[i * k + c] = pad[(size_t)i * 16 + c];
    return 0;
}

// V^T = kernel(Xs, X) L^-T, one row per new point (m x np, row-major): the same right-looking sweep as
// the factorisation's panel step, with the rows of kernel(Xs, X) in the role of the rows below the panel.
static int gs_check_series(gsum_ctx* ctx, const gsum_series_scale* sc) {
    if (sc->start < 0 || (sc->end >= 0 && sc->end < sc->start)) GS_FAIL("series scale: end must be >= start >= 0");
    if (sc->n_excluded < 0 || sc->n_excluded > GSUM_MAX_EXCLUDED) GS_FAIL("series scale: too many excluded orders");
    return 0;
}

int gsum_mat_scale_series(gsum_ctx* ctx, gsum_mat* A, const gsum_series_scale* sc, const double* ref, const double* ratio) {
    if (!ctx || !A || !sc || !ref || !ratio) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (A->factored) GS_FAIL("scale_series needs an unfactored matrix");
    if (gs_check_series(ctx, sc)) return -2;
    const int64_t n = A->n;
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, (size_t)2 * n * 8)) return -1;
    double* dref = ctx->scratch;
    double* drat = dref + n;
    hipStream_t s = ctx->cur->sm;
    GS_CHECK(hipMemcpyAsync(dref, ref, (size_t)n * 8, hipMemcpyHostToDevice, s));
    GS_CHECK(hipMemcpyAsync(drat, ratio, (size_t)n * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_scale_series, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, s, A->A, A->ld, (int)n, (int)n,
                       dref, drat, dref, drat, *sc);
    GS_CHECK(hipGetLastError());
    GS_CHECK(hipStreamSynchronize(s));
    return 0;
}

static int gs_predict_terms(gsum_ctx* ctx, gsum_mat* L, const gsum_kernel_desc* desc, const double* X, int64_t n,
                            int32_t d, const double* Xs, int64_t m, const double* RHS, int32_t k,
                            const gsum_series_scale* sc, const double* ref_x, const double* ratio_x,
                            const double* ref_s, const double* ratio_s, double* colsumsq, double* VtW, double* cov_out);

int gsum_predict_terms(gsum_ctx* ctx, gsum_mat* L, const gsum_kernel_desc* desc, const double* X, int64_t n,
                       int32_t d, const double* Xs, int64_t m, const double* RHS, int32_t k,
                       double* colsumsq, double* VtW, double* cov_out) {
    return gs_predict_terms(ctx, L, desc, X, n, d, Xs, m, RHS, k, nullptr, nullptr, nullptr, nullptr, nullptr, colsumsq,
                            VtW, cov_out);
}

int gsum_predict_terms_series(gsum_ctx* ctx, gsum_mat* L, const gsum_kernel_desc* desc, const double* X, int64_t n,
                              int32_t d, const double* Xs, int64_t m, const double* RHS, int32_t k,
                              const gsum_series_scale* sc, const double* ref_x, const double* ratio_x,
                              const double* ref_s, const double* ratio_s, double* colsumsq, double* VtW,
                              double* cov_out) {
    if (!ctx || !sc || !ref_x || !ratio_x || !ref_s || !ratio_s) return -2;
    if (gs_check_series(ctx, sc)) return -2;
    return gs_predict_terms(ctx, L, desc, X, n, d, Xs, m, RHS, k, sc, ref_x, ratio_x, ref_s, ratio_s, colsumsq, VtW,
                            cov_out);
}

static int gs_predict_terms(gsum_ctx* ctx, gsum_mat* L, const gsum_kernel_desc* desc, const double* X, int64_t n,
                            int32_t d, const double* Xs, int64_t m, const double* RHS, int32_t k,
                            const gsum_series_scale* sc, const double* ref_x, const double* ratio_x,
                            const double* ref_s, const double* ratio_s, double* colsumsq, double* VtW, double* cov_out) {
    if (!ctx || !L || !X || !Xs || !colsumsq) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (gs_check_desc(ctx, desc, d)) return -2;
    if (!L->factored) GS_FAIL("predict_terms needs a factorised matrix");
    if (n != L->n || m <= 0) GS_FAIL("bad shapes");
    if (k < 0 || k > GSUM_MAX_RHS || (k > 0 && (!RHS || !VtW))) GS_FAIL("bad RHS / k");
    const int64_t np = L->np, ld = L->ld, ldb = np + GS_BORDER;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t o_xs = 0, o_bt = up((size_t)m * d * 8), o_vw = o_bt + up((size_t)m * ldb * 8),
                 o_ss = o_vw + up((size_t)m * 16 * 8), o_cv = o_ss + up((size_t)m * 8),
                 o_sc = o_cv + (cov_out ? up((size_t)m * m * 8) : 0),
                 total = o_sc + (sc ? up((size_t)2 * (n + m) * 8) : 0);
    if (gs_upload_X(ctx, &ctx->op, X, n, d)) return -1;
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, total)) return -1;
    char* base = (char*)ctx->scratch;
    double *dXs = (double*)(base + o_xs), *Bt = (double*)(base + o_bt), *dVW = (double*)(base + o_vw),
           *dSS = (double*)(base + o_ss), *dCov = (double*)(base + o_cv);
    GS_CHECK(hipMemcpyAsync(dXs, Xs, (size_t)m * d * 8, hipMemcpyHostToDevice, ctx->cur->sm));
    if (gs_launch_build<true>(ctx, ctx->cur->sm, Bt, ldb, dXs, ctx->op.X, m, n, m, np, d, desc, 0.0, 0)) return -1;
    if (sc) {
        // rows of Bt are the new points, columns the conditioning points
        double* v = (double*)(base + o_sc);
        double *d_ref_s = v, *d_rat_s = v + m, *d_ref_x = v + 2 * m, *d_rat_x = v + 2 * m + n;
        GS_CHECK(hipMemcpyAsync(d_ref_s, ref_s, (size_t)m * 8, hipMemcpyHostToDevice, ctx->cur->sm));
        GS_CHECK(hipMemcpyAsync(d_rat_s, ratio_s, (size_t)m * 8, hipMemcpyHostToDevice, ctx->cur->sm));
        GS_CHECK(hipMemcpyAsync(d_ref_x, ref_x, (size_t)n * 8, hipMemcpyHostToDevice, ctx->cur->sm));
        GS_CHECK(hipMemcpyAsync(d_rat_x, ratio_x, (size_t)n * 8, hipMemcpyHostToDevice, ctx->cur->sm));
        hipLaunchKernelGGL(k_scale_series, dim3((unsigned)((n + 255) / 256), (unsigned)m), dim3(256), 0, ctx->cur->sm, Bt, ldb, (int)m,
                           (int)n, d_ref_s, d_rat_s, d_ref_x, d_rat_x, *sc);
        GS_CHECK(hipGetLastError());
    }
    // V^T = kernel(Xs, X) L^-T by a right-looking sweep, two block columns per trailing update (K = 256) like the
    // factorisation: the trailing part of Bt is read and written once per 256 eliminated columns instead of once per 128
    // (at m = 2048, n = 16384 a K = 128 sweep moved 0.5 GB per step against 190 us of MFMA work)
    const int sib_cfg = m >= 1024 ? GS_BULK : 1;
    bool deferred = false;                               // the columns right of the next panel still owe the previous panel's update
    for (int c = 0; c < L->T; c += 2) {
        const bool two = c + 1 < L->T;
        const int64_t c0 = (int64_t)c * GS_NB, c1 = c0 + GS_NB, r2 = two ? c1 + GS_NB : c1;
        if (gs_trsm_rows(ctx, ctx->cur->sm, L, c, Bt + c0, ldb, m)) return -1;
        if (two) {
            if (gs_gemm(ctx, ctx->cur->sm, sib_cfg, Bt + c1, ldb, Bt + c0, ldb, L->A + c1 * ld + c0, ld, m, GS_NB, GS_NB, 0, 1, -1.0))
                return -1;
            if (gs_trsm_rows(ctx, ctx->cur->sm, L, c + 1, Bt + c1, ldb, m)) return -1;
        }
        if (r2 >= np) continue;
        // The batch factorisation's lazy far updates (lazy_far = 2) applied to this sweep: after an even step only the next panel's 256 columns take
        // this panel's update (K = 256); the step after it applies both panels to everything to its right in ONE K = 512 launch -- half as many passes
        // over the trailing part of Bt, each at the tile kernel's better K = 512 rate.  Same products in the same ascending-k order per element.
        const bool pair = ctx->predict_lazy && two && m >= 1024 && np >= ctx->lazy_min_np && c + 3 < L->T && r2 + 2 * GS_NB <= np;
        if (!deferred && pair) {
            if (gs_gemm(ctx, ctx->cur->sm, GS_BULK, Bt + r2, ldb, Bt + c0, ldb, L->A + r2 * ld + c0, ld, m, 2 * GS_NB, (int)(r2 - c0), 0, 1, -1.0)) return -1;
            deferred = true;
        } else if (deferred) {
            const int64_t cp = c0 - 2 * GS_NB;            // the previous panel's first column: [cp, r2) is 512 columns wide
            if (gs_gemm(ctx, ctx->cur->sm, GS_BULK, Bt + r2, ldb, Bt + cp, ldb, L->A + r2 * ld + cp, ld, m, np - r2, (int)(r2 - cp), 0, 1, -1.0)) return -1;
            deferred = false;
        } else if (gs_gemm(ctx, ctx->cur->sm, GS_BULK, Bt + r2, ldb, Bt + c0, ldb, L->A + r2 * ld + c0, ld, m, np - r2, (int)(r2 - c0), 0, 1, -1.0))
            return -1;
    }
    std::vector<double> vw;
    if (k > 0) {
        // row sums of squares and V^T W in ONE pass over V^T (k_rowsumsq_vw)
        if (gs_border_prepare(ctx, L, RHS, n, k)) return -1;
        hipLaunchKernelGGL(k_rowsumsq_vw, dim3((unsigned)((m + 4 * GS_VW_ROWS - 1) / (4 * GS_VW_ROWS))), dim3(256), 0, ctx->cur->sm, Bt, ldb, (int)m, (int)np,
                           L->A + np * ld, ld, dSS, dVW);
        GS_CHECK(hipGetLastError());
        vw.resize((size_t)m * 16);
        GS_CHECK(hipMemcpyAsync(vw.data(), dVW, (size_t)m * 16 * 8, hipMemcpyDeviceToHost, ctx->cur->sm));
    } else {
        hipLaunchKernelGGL(k_rowsumsq, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, ctx->cur->sm, Bt, ldb, (int)m, (int)np, dSS);
        GS_CHECK(hipGetLastError());
    }
    GS_CHECK(hipMemcpyAsync(colsumsq, dSS, (size_t)m * 8, hipMemcpyDeviceToHost, ctx->cur->sm));
    if (cov_out) {
        // V^T V is symmetric: lower tiles only (half the flops of the square product), then mirrored in place
        if (gs_gemm(ctx, ctx->cur->sm, GS_BULK, dCov, m, Bt, ldb, Bt, ldb, m, m, (int)np, 1, 0, 1.0)) return -1;
        hipLaunchKernelGGL(k_mirror_lower, dim3((unsigned)((m + 255) / 256), (unsigned)m), dim3(256), 0, ctx->cur->sm, dCov, m, (int)m);
        GS_CHECK(hipGetLastError());
        GS_CHECK(hipMemcpyAsync(cov_out, dCov, (size_t)m * m * 8, hipMemcpyDeviceToHost, ctx->cur->sm));
    }
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    for (int64_t j = 0; j < m && k > 0; ++j)
        for (int c = 0; c < k; ++c) VtW[j * k + c] = vw[(size_t)j * 16 + c];
    return 0;
}

int gsum_mat_to_host(gsum_ctx* ctx, const gsum_mat* A, double* out) {
    if (!ctx || !A || !out) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    const int64_t n = A->n;
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, (size_t)n * n * sizeof(double))) return -1;
    hipLaunchKernelGGL(k_export, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, ctx->cur->sm, A->A, A->ld, (int)n,
                       ctx->scratch, A->factored ? 1 : 0);
    GS_CHECK(hipGetLastError());
    GS_CHECK(hipMemcpyAsync(out, ctx->scratch, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, ctx->cur->sm));
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    return 0;
}

int64_t gsum_mat_n(const gsum_mat* A) { return A ? A->n : -1; }

void gsum_mat_free(gsum_ctx* ctx, gsum_mat* A) {
    if (!A) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->cur->sm);
        if (ctx->cur->sp) (void)hipStreamSynchronize(ctx->cur->sp);
    }
    gs_mat_release(A);
}

int gsum_set_inputs(gsum_ctx* ctx, const double* X, int64_t n, int32_t d, const double* RHS, int32_t k) {
    if (!ctx) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    ctx->cur = &ctx->slots[0];
    if (d < 1 || d > GSUM_MAX_D) GS_FAIL("input dimension must be 1..GSUM_MAX_D");
    if (gs_upload_X(ctx, &ctx->res, X, n, d)) return -1;
    if (gs_upload_Z(ctx, &ctx->res, RHS, n, k)) return -1;
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    return 0;
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
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    return 0;
}

// enqueue one evaluation on the current slot (asynchronous: nothing waits on the host)
static int gs_eval_enqueue(gsum_ctx* ctx, const gsum_kernel_desc* desc, double nugget) {
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
    if (gs_set_border(ctx, sl->sm, m, ctx->in->Z, ctx->in->k)) return -1;
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
        const int rc = gs_eval_enqueue(ctx, &d, sl->last_nugget);
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
    const size_t o_desc = 0, o_res = up((size_t)CH * sizeof(gsum_kernel_desc)), o_scr = o_res + up((size_t)CH * 258 * 8);
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, o_scr + (size_t)CH * GS_SMALL_SCRATCH * 8)) return -1;
    char* base = (char*)ctx->scratch;
    if (gs_reserve_pinned(ctx, (size_t)CH * 258 * 8)) return -1;
    double* hres = ctx->hbatch;
    for (int lo = 0; lo < n_kernels; lo += CH) {
        const int cnt = std::min(CH, n_kernels - lo);
        GS_CHECK(hipMemcpyAsync(base + o_desc, kernels + lo, (size_t)cnt * sizeof(gsum_kernel_desc), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_lml_small, dim3(cnt), dim3(256), 0, s, ctx->in->X, (int)ctx->in->n, ctx->in->d, ctx->in->Z, k,
                           (const gsum_kernel_desc*)(base + o_desc), nugget, (double*)(base + o_scr), (double*)(base + o_res));
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
    const int64_t n = ctx->in->n, np = (n + GS_NB - 1) / GS_NB * GS_NB, T = np / GS_NB, ld = np + GS_BORDER;
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
    const size_t o_desc = 0, o_res = up((size_t)CH * sizeof(gsum_kernel_desc)), o_scr = o_res + up((size_t)CH * 258 * 8);
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, o_scr + (size_t)CH * stride * 8)) return -1;
    char* base = (char*)ctx->scratch;
    const size_t shmem = (size_t)std::max<int>(GS_TILE_LD_DOUBLES, GS_DIAG_WS) * sizeof(double);
    if (!ctx->lds_attr_done.count((const void*)k_lml_medium)) {
        GS_CHECK(hipFuncSetAttribute((const void*)k_lml_medium, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        ctx->lds_attr_done.insert((const void*)k_lml_medium);
    }
    if (gs_reserve_pinned(ctx, (size_t)CH * 258 * 8)) return -1;
    double* hres = ctx->hbatch;
    for (int lo = 0; lo < n_kernels; lo += CH) {
        const int cnt = std::min(CH, n_kernels - lo);
        GS_CHECK(hipMemcpyAsync(base + o_desc, kernels + lo, (size_t)cnt * sizeof(gsum_kernel_desc), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_lml_medium, dim3(cnt), dim3(256), shmem, s, ctx->in->X, (int)n, ctx->in->d, ctx->in->Z, k,
                           (const gsum_kernel_desc*)(base + o_desc), nugget, (double*)(base + o_scr), stride, (double*)(base + o_res),
                           ctx->diag_stamps ? ctx->dstamps : (unsigned long long*)nullptr);
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
    return (double)(np + GS_BORDER) * (double)(np + GS_BORDER) * 8.0 + T * GS_LTAB * 8.0 + (T / 2 + 1) * GS_LSIB * 8.0 +
           (T + (double)np + 258.0) * 8.0 + 4.0;
}

static int gs_wave_prepare(gsum_ctx* ctx, int G, int B, int64_t n, int64_t np) {
    gs_wave* wv = &ctx->wave;
    if (!wv->sb) wv->sb = ctx->slots[0].sm;
    const int T = (int)(np / GS_NB);
    const int64_t ld = np + GS_BORDER;
    for (int i = 0; i < G; ++i) {
        gs_wave_group* g = &wv->g[i];
        if (!g->sc) {
            // A batch call and a single factorisation never run at the same time: the first two groups run on slot 0's two
            // high-priority streams, the third on the stream gsum_init created next to them (see there: four streams on four pipes)
            gs_slot* s0 = &ctx->slots[0];
            if (i == 0) { if (gs_panel_stream(ctx, s0)) return -1; g->sc = s0->sp; }
            else if (i == 1) { if (gs_aux_stream(ctx, s0)) return -1; g->sc = s0->sa; }
            else { GS_CHECK(hipStreamCreateWithPriority(&g->sc, hipStreamNonBlocking, ctx->prio_hi)); g->own_sc = true; }
        }
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

static int gs_lml_wave(gsum_ctx* ctx, const gsum_kernel_desc* kernels, int n_kernels, double nugget, double* G_out, double* sld_out,
                       int64_t* info_out) {
    const int64_t n = ctx->in->n, np = (n + 2 * GS_NB - 1) / (2 * GS_NB) * (2 * GS_NB), ld = np + GS_BORDER, naug = np + GS_BORDER;
    const int k = ctx->in->k, d = ctx->in->d, S = (int)(np / (2 * GS_NB));
    int G = std::max(1, std::min(GS_WV_GROUPS, ctx->wave_groups));
    int B = std::max(1, std::min(GS_WVC_MAX, ctx->wave_size));
    if (n_kernels < G * B) {                      // a short call: every evaluation in flight at once, the groups equally full
        G = std::min(G, n_kernels);
        B = (n_kernels + G - 1) / G;
    }
    {
        // group workspaces within 70 % of what is free (plus what the groups already hold)
        size_t free_b = 0, total_b = 0;
        GS_CHECK(hipMemGetInfo(&free_b, &total_b));
        double held = 0.0;
        for (int i = 0; i < GS_WV_GROUPS; ++i)
            if (ctx->wave.g[i].cap && ctx->wave.g[i].n == n) held += ctx->wave.g[i].cap * gs_wave_ws_bytes(np);
        const int fit = (int)std::min<double>(1e6, (0.7 * (double)free_b + held) / gs_wave_ws_bytes(np));
        if (fit < 1) GS_FAIL("not enough device memory for one workspace matrix");
        if (G * B > fit) {
            G = std::max(1, std::min(G, fit));
            B = std::max(1, fit / G);
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
    if (gs_wave_prepare(ctx, G, B, n, np)) return -1;
    ctx->wave_last_streams = G + 1;
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
    const int shift = !several_rounds ? 0 : (ctx->wave_shift > 0 ? std::min(ctx->wave_shift, S) : 0);
    for (int i = 0; i < G; ++i) {
        gs_wave_group* g = &wv->g[i];
        g->active = false;
        g->cnt = g->step = 0;
        g->start_tick = i * shift;
    }
    const bool prof = ctx->profile_gemm > 0;
    if (prof) ctx->prof_this_eval = true;
    // everything of this call follows what the context's main stream has done so far (the resident inputs' upload)
    hipStream_t s0 = ctx->slots[0].sm;
    GS_CHECK(hipEventRecord(ctx->slots[0].evFork, s0));
    GS_CHECK(hipStreamWaitEvent(wv->sb, ctx->slots[0].evFork, 0));
    for (int i = 0; i < G; ++i) GS_CHECK(hipStreamWaitEvent(wv->g[i].sc, ctx->slots[0].evFork, 0));
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
                    const int rec = gs_prof_begin(ctx, g->sc, GS_PROF_BUILD, 0.0);
                    const int rc = gs_launch_build<false>(ctx, g->sc, g->pool.A + (int64_t)e * g->pool.strideA, ld, ctx->in->X, nullptr, n, n,
                                                          np, np, d, &kernels[next + e], nugget, ctx->build_lower_only);
                    gs_prof_end(ctx, g->sc, rec);
                    if (rc) return rc;
                }
                next += g->cnt;
                gs_wave_fill_chain(g, &ca, false);
                const int rec = gs_prof_begin(ctx, g->sc, GS_PROF_OTHER, 0.0);
                hipLaunchKernelGGL(k_set_border_g, dim3((unsigned)((naug + 255) / 256), (unsigned)g->cnt), dim3(256), 0, g->sc, ca, (int)n,
                                   (const double*)ctx->in->Z, k);
                hipLaunchKernelGGL(k_wave_begin, dim3((unsigned)((np + 255) / 256), (unsigned)g->cnt), dim3(256), 0, g->sc, ca);
                gs_prof_end(ctx, g->sc, rec);
                GS_CHECK(hipGetLastError());
            } else {
                GS_CHECK(hipStreamWaitEvent(g->sc, g->evBulk, 0));            // the trailing update of the previous step
            }
            // ---- one macro-step: the chain of outer step g->step (diagonal super-blocks, then both panels of all rows below them) and
            // its trailing update.  A "near" update (the next panel's 256 columns only, K = 256: ~1 GF per member) sits on the chain's
            // critical path -- chain(s) -> near(s) -> chain(s + 1) -- and goes out on the CHAIN stream, followed at once by the next
            // step's chain; only the big updates (whole lower triangle, K = 512 or 256) go to the bulk stream.  So between two of its
            // big updates a group needs diag + panel + near + diag + panel (~0.8 ms) and the other groups' big updates cover it.
            for (;;) {
                {
                    const int rec = gs_prof_begin(ctx, g->sc, GS_PROF_DIAG, (double)g->cnt * 8.0 * GS_NB * GS_NB * GS_NB / 3.0);
                    gs_wave_fill_chain(g, &ca, false);
                    hipLaunchKernelGGL(k_potrf_diag256g, dim3((unsigned)g->cnt), dim3(256), 0, g->sc, ca);
                    gs_prof_end(ctx, g->sc, rec);
                }
                const int64_t c0 = 2 * GS_NB * (int64_t)g->step, r2 = c0 + 2 * GS_NB, mrest = naug - r2;
                const bool serial = ctx->wave_serial != 0;
                if (serial) {                  // only the diagonal blocks run beside the bulk stream's kernels (see wave_serial)
                    GS_CHECK(hipEventRecord(g->evChain, g->sc));
                    GS_CHECK(hipStreamWaitEvent(wv->sb, g->evChain, 0));
                }
                {
                    hipStream_t spn = serial ? wv->sb : g->sc;
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
                const bool near = st.near && ctx->wave_near_on_chain && !serial;
                hipStream_t su = near ? g->sc : wv->sb;
                if (!near && !serial) {
                    GS_CHECK(hipEventRecord(g->evChain, g->sc));
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
                const int tiles = (int)(en.tri ? tm * (tm + 1) : tm * ((en.N + 63) / 64));
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
                const size_t shm = 2 * (size_t)((128 + 64) * GS_KC + 4) * sizeof(double);
                if (near) hipLaunchKernelGGL(k_gemm_ld3n, dim3((unsigned)run), dim3(512), shm, su, ga);
                else hipLaunchKernelGGL(k_gemm_ld3g, dim3((unsigned)run), dim3(512), shm, su, ga);
                gs_prof_end(ctx, su, rec);
                GS_CHECK(hipGetLastError());
                ++g->step;
                if (!near) break;
            }
            GS_CHECK(hipEventRecord(g->evBulk, wv->sb));
            if (g->step < S) continue;
            // ---- the round is complete: read-out on the chain stream (the bulk stream goes on with the other groups)
            GS_CHECK(hipStreamWaitEvent(g->sc, g->evBulk, 0));
            gs_wave_fill_chain(g, &ca, false);
            const int rec = gs_prof_begin(ctx, g->sc, GS_PROF_OTHER, 0.0);
            hipLaunchKernelGGL(k_finalize_g, dim3((unsigned)g->cnt), dim3(256), 0, g->sc, ca);
            gs_prof_end(ctx, g->sc, rec);
            GS_CHECK(hipGetLastError());
            GS_CHECK(hipMemcpyAsync(ctx->hbatch + (size_t)g->first_eval * 258, g->pool.res, (size_t)g->cnt * 258 * sizeof(double),
                                    hipMemcpyDeviceToHost, g->sc));
            g->active = false;
            --live;
        }
    }
    for (int i = 0; i < G; ++i) GS_CHECK(hipStreamSynchronize(wv->g[i].sc));
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

static int gs_lml_on(gsum_ctx* ctx, gs_inputs* I, const gsum_kernel_desc* kernels, int32_t n_kernels, double nugget,
                     double* G_out, double* sld_out, int64_t* info_out) {
    if (!ctx || !kernels || !G_out || !sld_out || !info_out) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    ctx->in = I;
    if (!ctx->in->X) GS_FAIL("gsum_set_inputs has not been called");
    for (int i = 0; i < n_kernels; ++i)
        if (gs_check_desc(ctx, &kernels[i], ctx->in->d)) return -2;
    bool any_tree = false;                       // the one-workgroup-per-evaluation kernels build the flattened form only
    for (int i = 0; i < n_kernels; ++i) any_tree = any_tree || kernels[i].n_ops > 0;
    if (ctx->in->n <= GS_NB && ctx->small_path && !any_tree) {
        ctx->cur = &ctx->slots[0];
        return gs_lml_small(ctx, kernels, n_kernels, nugget, G_out, sld_out, info_out);
    }
    // break-even against the grouped schedule, re-measured in round 4 (the grouped launches made small batches much faster than round 3's
    // 20 streams; tools/gpu_medium_breakeven.py, profiles/r04_medium_breakeven.log): the fused path wins from 2, ~24, ~56, ~104, ~130, ~190,
    // ~215 evaluations at n = 256, 512, 1024, 1536, 2048, 3072, 4096 -- n / 16 above n = 256 (round 3's rule n^1.55 / 2000 chose the fused
    // path up to 40 % too early: n = 2048, 96 evaluations 17.5 ms fused against 12.4 grouped)
    const int med_min = ctx->medium_min_batch > 0 ? ctx->medium_min_batch
                                                  : (ctx->in->n <= 256 ? 2 : std::max(4, (int)(ctx->in->n / 16)));
    if (ctx->in->n <= GS_MEDIUM_MAX && ctx->medium_path && n_kernels >= med_min && !any_tree) {
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
        rc = gs_eval_enqueue(ctx, &kernels[i], nugget);
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
static int gs_grad_enqueue(gsum_ctx* ctx, gs_slot* sl, const gsum_kernel_desc* desc, const gsum_grad_param* params, int P, double nugget,
                           bool solo) {
    const int64_t n = ctx->in->n;
    const int d = ctx->in->d;
    ctx->cur = sl;
    ctx->chain_events_needed = 1;          // the sweep below trails the factorisation by its evP events (host-enqueued schedule)
    const int rc_eval = gs_eval_enqueue(ctx, desc, nugget);
    ctx->chain_events_needed = 0;
    if (rc_eval) return -1;
    gsum_mat* m = sl->ws;
    const int64_t np = m->np, ld = m->ld, ldg = np + GS_BORDER;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const int chunks = (int)std::min<int64_t>(128, (n + 63) / 64), rows_per = (int)((n + chunks - 1) / chunks);
    const size_t o_u = 0, o_r = up((size_t)np * ldg * 8), o_v = o_r + up((size_t)np * ldg * 8), o_q = o_v + up((size_t)16 * ldg * 8),
                 o_t = o_q + up((size_t)P * n * 16 * 8), o_o = o_t + up((size_t)P * n * 8), o_p = o_o + up((size_t)P * 257 * 8),
                 total = o_p + up((size_t)P * chunks * 257 * 8);
    if (gs_reserve(ctx, &sl->gws, &sl->gws_cap, total)) return -1;
    if (!sl->hgrad) GS_CHECK(hipHostMalloc((void**)&sl->hgrad, (size_t)GSUM_MAX_GRAD * 257 * sizeof(double), hipHostMallocDefault));
    char* base = (char*)sl->gws;
    double *U = (double*)(base + o_u), *Ri = (double*)(base + o_r), *Vt = (double*)(base + o_v), *Q = (double*)(base + o_q),
           *trow = (double*)(base + o_t), *dout = (double*)(base + o_o), *part = (double*)(base + o_p);
    hipStream_t s = sl->sm;
    // U = L^-T.  Solo: on a stream of its own, trailing the factorisation: block columns c, c + 1 of the sweep need the factor's
    // columns c0 .. c0 + 255 and their tables, which are final once the panel chain of that outer step has run (event
    // evP[c] of the look-ahead schedule).  One factorisation alone is bound by its panel chain, with most of the chip idle
    // behind it -- the sweep's GEMMs (n^3 / 3 flops) fill that time instead of following it (5.5 ms at n = 8192).
    const bool trail = solo && ctx->lookahead != 0 && ctx->batch_active < 3;       // the condition under which gs_potrf records evP
    hipStream_t su = s;
    if (solo) {
        if (!sl->su) {
            // the sweep runs beside the factorisation's main and panel streams: it takes the context's fourth stream (the third group's chain
            // stream of a batch, idle here) -- a stream created now would share a command-processor pipe with one of those two
            // (round 4 found the single gradient evaluation at 19.7 ms instead of 14.2 that way)
            if (sl == &ctx->slots[0] && ctx->wave.g[2].sc) { sl->su = ctx->wave.g[2].sc; sl->own_su = false; }
            else GS_CHECK(hipStreamCreateWithPriority(&sl->su, hipStreamNonBlocking, ctx->prio_lo));
            GS_CHECK(hipEventCreateWithFlags(&sl->evU, hipEventDisableTiming));
        }
        su = sl->su;
        GS_CHECK(hipEventRecord(sl->evU, s));                                  // everything enqueued so far (nothing of U is in use)
        if (!trail) GS_CHECK(hipStreamWaitEvent(su, sl->evU, 0));              // no per-panel events: the sweep follows the factorisation
    }
    hipLaunchKernelGGL(k_set_identity, dim3((unsigned)((np + 255) / 256), (unsigned)np), dim3(256), 0, su, U, ldg, (int)np);
    GS_CHECK(hipGetLastError());
    // two block columns per trailing update (K = 256), like the factorisation: halves the traffic of U's trailing part
    for (int c = 0; c < m->T; c += 2) {
        const bool two = c + 1 < m->T;
        const int64_t c0 = (int64_t)c * GS_NB, c1 = c0 + GS_NB, r2 = two ? c1 + GS_NB : c1;
        if (trail) GS_CHECK(hipStreamWaitEvent(su, sl->evP[c], 0));
        if (gs_trsm_rows(ctx, su, m, c, U + c0, ldg, c1)) return -1;
        if (two) {
            // rows below c1 are still zero in block column c: only rows < c1 feed the sibling column
            if (gs_gemm(ctx, su, 1, U + c1, ldg, U + c0, ldg, m->A + c1 * ld + c0, ld, c1, GS_NB, GS_NB, 0, 1, -1.0)) return -1;
            if (gs_trsm_rows(ctx, su, m, c + 1, U + c1, ldg, r2)) return -1;
        }
        if (r2 < np && gs_gemm(ctx, su, GS_BULK, U + r2, ldg, U + c0, ldg, m->A + r2 * ld + c0, ld, r2, np - r2, (int)(r2 - c0), 0, 1, -1.0))
            return -1;
    }
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
    hipLaunchKernelGGL(k_upper_times_rows, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, sv, U, ldg, (int)np, m->A + np * ld, ld, Vt, ldg);
    GS_CHECK(hipGetLastError());
    if (solo) {
        if (gs_potrf_events(ctx, sl, 1)) return -1;
        GS_CHECK(hipEventRecord(sl->evP[0], sl->sp));
    }
    if (gs_gemm(ctx, s, GS_BULK, Ri, ldg, U, ldg, U, ldg, np, np, (int)np, 2, 0, 1.0)) return -1;
    if (solo) GS_CHECK(hipStreamWaitEvent(s, sl->evP[0], 0));
    gs_grad_params prm;
    memset(&prm, 0, sizeof prm);
    for (int p = 0; p < P; ++p) prm.p[p] = params[p];
    if (desc->n_ops > 0) {
        hipLaunchKernelGGL(k_grad_contract<true>, dim3((unsigned)((n + 3) / 4), (unsigned)P), dim3(256), 0, s, ctx->in->X, (int)n, (int)d, *desc,
                       prm, Ri, ldg, Vt, ldg, Q, trow);
    } else {
        hipLaunchKernelGGL(k_grad_contract<false>, dim3((unsigned)((n + 3) / 4), (unsigned)P), dim3(256), 0, s, ctx->in->X, (int)n, (int)d, *desc,
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

int gsum_lml_grad(gsum_ctx* ctx, const gsum_kernel_desc* desc, const gsum_grad_param* params, int32_t n_params,
                  const double* X, int64_t n, int32_t d, const double* RHS, int32_t k, double nugget, double* G_out,
                  double* sld_out, int64_t* info_out, double* trace_out, double* H_out) {
    if (!ctx || !desc || !params || !G_out || !sld_out || !info_out || !trace_out || !H_out) return -2;
    if (gs_grad_check(ctx, params, n_params, d, k)) return -2;
    int rc = gs_upload_inputs(ctx, X, n, d, RHS, k);
    if (rc) return rc;
    ctx->in = &ctx->op;
    if (gs_check_desc(ctx, desc, d)) return -2;
    gs_slot* sl = &ctx->slots[0];
    ctx->batch_active = 1;
    if (gs_grad_enqueue(ctx, sl, desc, params, n_params, nugget, true)) return -1;
    sl->pending = 0;
    return gs_grad_harvest(ctx, sl, n_params, G_out, sld_out, info_out, trace_out, H_out);
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
    int rc = gs_upload_inputs(ctx, X, n, d, RHS, k);
    if (rc) return rc;
    ctx->in = &ctx->op;
    for (int i = 0; i < n_desc; ++i)
        if (gs_check_desc(ctx, &descs[i], d)) return -2;
    if (n_desc == 1) {
        gs_slot* sl = &ctx->slots[0];
        ctx->batch_active = 1;
        if (gs_grad_enqueue(ctx, sl, &descs[0], params, n_params, nugget, true)) return -1;
        sl->pending = 0;
        return gs_grad_harvest(ctx, sl, n_params, G_out, sld_out, info_out, trace_out, H_out);
    }
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

}  // extern "C"
