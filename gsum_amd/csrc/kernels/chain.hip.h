// the chain of a factorisation: flags, k_potrf_diag256 / k_panel256 (+ grouped forms), the persistent chain kernel k_chain
// (part of gsum_kernels.hip.h: included from there, in order; gfx950 only)
#pragma once
// ---- flags of the persistent-chain schedule (the schedule itself: after k_potrf_diag256, below) ----
#define GS_CH_GMAX 32                       // window row groups at most (W = 512)
#define GS_FL_ABORT 0                       // 0 running, 1 a wait timed out, 2 a pivot failed (info says where)
#define GS_FL_RESIDENT 1                    // workgroups of k_chain that have started (k_wait_flag holds the other streams back)
#define GS_FL_BASE 16
enum { GS_FL_T0 = 0, GS_FL_TL, GS_FL_T1, GS_FL_WTOP, GS_FL_WALL, GS_FL_UD0, GS_FL_UD1, GS_FL_UR, GS_FL_FA, GS_FL_FB, GS_FL_RP, GS_FL_FF, GS_FL_KINDS };
// (FF[s]: tiles of the deep schedule's far launch at the end of the macro-step that closes with step s, over the columns of the NEXT macro-step's near band)
__host__ __device__ inline int gs_fl(int kind, int S, int s) { return GS_FL_BASE + kind * S + s; }
__host__ __device__ inline int gs_fl_wg(int S, int s, int g) { return GS_FL_BASE + GS_FL_KINDS * S + GS_CH_GMAX * s + g; }
__host__ __device__ inline int gs_fl_count(int S) { return (GS_FL_BASE + (GS_FL_KINDS + GS_CH_GMAX) * S + 3) / 4 * 4; }
#define GS_CH_TIMEOUT 100000000ull          // 1 s of s_memrealtime (100 MHz)
#define GS_CH_STAMPS 16                     // u64 per outer step (diagnostics)
#define GS_CH_KSTAMPS 8                     // ... and first start / last end of the step's four host-enqueued launches
#define GS_CH_LDS_DOUBLES GS_LSIB           // 128 KB: the L10 operand images (>= the diagonal routine's 75.8 KB workspace)

#define GS_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ unsigned gs_flag_ld(const unsigned* f) {
    return (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(f, GS_RLX_AGENT));
}
__device__ __forceinline__ void gs_flag_st(unsigned* f, unsigned v) { __hip_atomic_store(f, v, GS_RLX_AGENT); }
__device__ __forceinline__ void gs_flag_add(unsigned* f) { (void)__hip_atomic_fetch_add(f, 1u, GS_RLX_AGENT); }
__device__ __forceinline__ void gs_st_wt(double* p, double v) { __hip_atomic_store(p, v, GS_RLX_AGENT); }   // global_store_dwordx2 sc1
__device__ __forceinline__ void gs_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// One wave polls until *f >= want (every lane loads the same word: one request).  false: the chain was aborted.  No acquire.
__device__ __forceinline__ bool gs_poll_ge(const unsigned* f, unsigned want, unsigned* flags) {
    if (gs_flag_ld(f) < want) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned spins = 0;; ++spins) {
            // short naps first (a hand-off on the critical path), longer ones once the wait is clearly a long one (the chain
            // idling behind the bulk update in the first third of a factorisation): polls are fabric traffic the bulk kernels pay for
            if (spins < 64) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(16);
            if (gs_flag_ld(f) >= want) break;
            if ((spins & 15) == 15) {
                if (gs_flag_ld(flags + GS_FL_ABORT)) return false;
                if (__builtin_amdgcn_s_memrealtime() - t0 > GS_CH_TIMEOUT) {
                    gs_flag_st(flags + GS_FL_ABORT, 1u);
                    return false;
                }
            }
        }
    }
    return true;
}
// ONE agent-scope acquire after the poll(s) have matched: this CU's L1 drops its lines; the wave's own later loads are ordered
// behind the invalidate in its memory pipeline (other waves: s_waitcnt vmcnt(0) + barrier first, gs_wg_wait_ge)
__device__ __forceinline__ void gs_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ bool gs_wait_ge(const unsigned* f, unsigned want, unsigned* flags) {
    if (!gs_poll_ge(f, want, flags)) return false;
    gs_acquire();
    return true;
}

// the same for a whole workgroup: thread 0's wave polls and acquires, the others load behind the barrier.  `sh`: one int of LDS.
__device__ __forceinline__ bool gs_wg_wait_ge(const unsigned* f, unsigned want, unsigned* flags, volatile int* sh) {
    if (threadIdx.x < 64) {
        const bool ok = gs_wait_ge(f, want, flags);
        gs_drain();                              // the invalidate has completed before the barrier lets the other waves load
        if (threadIdx.x == 0) *sh = ok ? 1 : 0;
    }
    __syncthreads();
    const int ok = *sh;
    __syncthreads();
    return ok != 0;
}

__global__ void k_signal(unsigned* f, unsigned v) {
    if (threadIdx.x == 0) gs_flag_st(f, v);
}

// one wave that waits for chain flags in stream order: everything enqueued behind it on its stream starts only then (the
// launch boundary is the acquire).  This is how the host-enqueued kernels of the schedule meet the chain: a poll + acquire +
// two barriers in front of EVERY workgroup of a 3000-workgroup trailing update cost 24 us per outer step (measured), and
// gated workgroups hold their slots while they spin; one spinning wave costs nothing.
// Also once per factorisation: nothing is dispatched before EVERY workgroup of k_chain is resident -- a k_chain wave needs a whole
// SIMD's registers and its workgroup most of a CU's LDS, and other streams' waves that wait for a chain workgroup that found no
// room would keep it out for good (seen: one factorisation in three timed out at n = 8192).
__global__ __launch_bounds__(64) void k_wait_flag(const unsigned* f, unsigned want, const unsigned* f2, unsigned want2, unsigned* flags) {
    if (gs_poll_ge(f, want, flags) && f2) (void)gs_poll_ge(f2, want2, flags);
}

// end of a persistent-chain factorisation: a chain that gave up (a wait timed out) says so through the info word
#define GS_INFO_CHAIN_ABORT 0x7fffffff
__global__ void k_chain_status(const unsigned* flags, int* info) {
    if (threadIdx.x == 0 && gs_flag_ld(flags + GS_FL_ABORT) == 1u) *info = GS_INFO_CHAIN_ABORT;
}

// two-stream probe of the chain schedule's one assumption: kernels of different streams of this process run side by side
// (a profiler that serialises dispatches breaks it).  k_probe_wait spins until k_signal's word arrives or `ticks` pass.
__global__ __launch_bounds__(64) void k_probe_wait(const unsigned* f, const unsigned* f2, unsigned long long ticks, unsigned* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned seen = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        if (gs_flag_ld(f) && gs_flag_ld(f2)) { seen = 1; break; }
        __builtin_amdgcn_s_sleep(8);
    }
    if (threadIdx.x == 0) *out = seen;
}

// rows [0, M) x 256 columns at P: both panels of an outer step in one launch, 16 rows per single-wave workgroup:
//   X_j = B_j L_jj^-T;   B_j+1 -= X_j L(j+1, j)^T;   X_j+1 = B_j+1 L_j+1,j+1^-T
// (k_panel, the K = 128 sibling update on k_gemm_nt, k_panel again -- without two launches and two passes over the rows)
__device__ __forceinline__ void gs_panel256_body(double* P, int64_t ld, int M, const double* Ltab0, const double* Lsib,
                                                 const double* Ltab1, unsigned long long* kst, unsigned long long* wstat, const int group,
                                                 double* tile = nullptr) {
    const int lane = threadIdx.x & 63;
    const int r0 = group * 16;
    if (r0 >= M) return;
    const unsigned long long w_t0 = wstat ? __builtin_amdgcn_s_memrealtime() : 0ull;
    __builtin_amdgcn_s_setprio(3);
    if (kst && lane == 0) atomicMin(kst, __builtin_amdgcn_s_memrealtime());          // diagnostics: first start / last end of the launch
    double* rows = P + (int64_t)r0 * ld;
    gs_d4 P0[8], P1[8];
    if (tile) gs_panel16_load_t(P0, rows, ld, M - r0, lane, tile); else gs_panel16_load(P0, rows, ld, M - r0, lane);
    gs_panel16_solve_g(P0, Ltab0, lane);
    if (tile) gs_panel16_store_t(P0, rows, ld, M - r0, lane, tile); else gs_panel16_store(P0, rows, ld, M - r0, lane);
    __builtin_amdgcn_sched_barrier(0);          // the second 128 columns are fetched only now: 64 registers less at the peak
    if (tile) gs_panel16_load_t(P1, rows + 128, ld, M - r0, lane, tile); else gs_panel16_load(P1, rows + 128, ld, M - r0, lane);
    gs_sib_update_lean(P1, P0, Lsib, lane);
    __builtin_amdgcn_sched_barrier(0);
    gs_panel16_solve_g(P1, Ltab1, lane);
    if (tile) gs_panel16_store_t(P1, rows + 128, ld, M - r0, lane, tile); else gs_panel16_store(P1, rows + 128, ld, M - r0, lane);
    if (kst && lane == 0) atomicMax(kst + 1, __builtin_amdgcn_s_memrealtime());
    if (wstat && lane == 0) {                                   // diagnostics (option panel_stats): how long the panel's waves are resident
        atomicAdd(wstat, __builtin_amdgcn_s_memrealtime() - w_t0);
        atomicAdd(wstat + 1, 1ull);
    }
}

__global__ __launch_bounds__(64, 2) void k_panel256(double* P, int64_t ld, int M, const double* Ltab0, const double* Lsib,
                                                  const double* Ltab1, unsigned long long* kst, unsigned long long* wstat) {
    __shared__ __attribute__((aligned(16))) double tile[GS_PT_TILE];
    gs_panel256_body(P, ld, M, Ltab0, Lsib, Ltab1, kst, wstat, (int)blockIdx.x, tile);
}

// explicit inverses of the diagonal blocks from their tables, one workgroup per block (for the consumers that still
// multiply by L_bb^-1: the back-substitution half of cho_solve)
__global__ __launch_bounds__(256) void k_trtri_blocks(const double* Ltab, double* Linv) {
    __shared__ __attribute__((aligned(16))) double tab[GS_LTAB];
    gs_load_ltab(tab, Ltab + (size_t)blockIdx.x * GS_LTAB);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    gs_trtri_block(tab + GS_D2_LS, tab + GS_D2_DV, Linv + (size_t)blockIdx.x * 128 * 128, w, lane);
}

// The operand images of the sibling blocks L(k + 1, k), k even, from the FACTOR in the matrix -- what k_potrf_diag256 / k_chain leave behind
// as a by-product (-L10 in gs_panel16_load's register layout: [row group][k-block][x][lane]); for consumers of k_panel256 on a factor whose
// schedule did not produce them (the unfused host-enqueued schedule of small orders): the predictive sweep.  One workgroup per pair.
__global__ __launch_bounds__(256) void k_make_lsib(const double* A, int64_t ld, double* Lsib) {
    const int s = blockIdx.x;
    const double* L10 = A + ((int64_t)(2 * s + 1) * GS_NB) * ld + (int64_t)(2 * s) * GS_NB;
    double* img = Lsib + (size_t)s * GS_LSIB;
    for (int i = threadIdx.x; i < 8 * 8 * 4 * 64; i += 256) {
        const int lane = i & 63, x = (i >> 6) & 3, kb = (i >> 8) & 7, c = i >> 11;
        const int fr = lane & 15, fq = lane >> 4;
        img[i] = -L10[(int64_t)(16 * c + fr) * ld + 16 * kb + fq + 4 * x];
    }
}

// info: global failure flag (0 = ok so far; >0 = LAPACK-style 1-based failing column).  Micro-block routine, substitution tables to Ltab.
__global__ __launch_bounds__(256) void k_potrf_diag(double* A, int64_t ld, double* Ltab, double* logdet, int* info, int col0,
                                                     const double* diag0, unsigned long long* stamps) {
    // 34 KB of LDS (one panel column at a time, dumps exported as they are made): with its 124 VGPRs the workgroup fits where ONE
    // bulk workgroup (53 KB, 8 waves) has just retired; at 77 KB it waited for two on the same CU while lower-priority bulk
    // workgroups kept taking the single slots (rocprofv3: 90-250 us per call beside the bulk update)
    __shared__ __attribute__((aligned(16))) double wsd[GS_D2C_WS];
    if (*info != 0) return;                    // an earlier block already failed (uniform)
    __builtin_amdgcn_s_setprio(3);             // the chain's one workgroup: ahead of the bulk waves on its SIMDs
    const int bad = gs_diag_block<false>(A, ld, (double*)nullptr, Ltab, logdet, diag0, stamps, wsd);
    if (bad && threadIdx.x == 0) *info = col0 + bad;
}

// Two diagonal blocks in one launch: the 256 x 256 diagonal super-block of an outer step, by one workgroup.
//   A00 = L00 L00^T (gs_diag_block);  L10 = A10 L00^-T (blocked substitution, two 16-row groups per wave);
//   A11 -= L10 L10^T (lower micro-tiles, on accumulators that start as the matrix entries, ascending k: k_gemm_nt's
//   arithmetic);  A11 = L11 L11^T (gs_diag_block).
// Replaces diag / panel / sibling update / diag on the chain of a factorisation: four dependent launches, two of them
// over all rows below, become one; the rows below go through k_panel256 afterwards.  L10 is also left in Lsib (operand
// layout) for that kernel.  Tables of both blocks to Ltab[0], Ltab[GS_LTAB].
__device__ __forceinline__ void gs_potrf_diag256_body(double* A, int64_t ld, double* Ltab, double* Lsib, double* logdet, int* info,
                                                      int col0, const double* diag0, unsigned long long* stamps, double* wsd) {
    if (*info != 0) return;
    __builtin_amdgcn_s_setprio(3);
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    int bad = gs_diag_block<false>(A, ld, (double*)nullptr, Ltab, logdet, diag0, stamps, wsd);
    if (bad) {
        if (t == 0) *info = col0 + bad;
        return;
    }
    __threadfence();                            // the tables just written are read back from global memory below
    __syncthreads();
    // ---- L10: groups w and 7 - w of the 128 rows below
    double* A10 = A + (int64_t)128 * ld;
    const int g0 = w, g1 = 7 - w;
    gs_d4 Pa[8], Pb[8];
    gs_panel16_load(Pa, A10 + (int64_t)(16 * g0) * ld, ld, 16, lane);
    gs_panel16_load(Pb, A10 + (int64_t)(16 * g1) * ld, ld, 16, lane);
    gs_panel16_solve_g(Pa, Ltab, lane);
    gs_panel16_solve_g(Pb, Ltab, lane);
    gs_panel16_store(Pa, A10 + (int64_t)(16 * g0) * ld, ld, 16, lane);
    gs_panel16_store(Pb, A10 + (int64_t)(16 * g1) * ld, ld, 16, lane);
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            Lsib[((g0 * 8 + k) * 4 + x) * 64 + lane] = Pa[k][x];
            Lsib[((g1 * 8 + k) * 4 + x) * 64 + lane] = Pb[k][x];
        }
    __threadfence();
    __syncthreads();
    // ---- A11 -= L10 L10^T: micro-tile (c, c') for c' in {g0, g1}, c >= c'.  A operand: dump of group c (-L_c,kb),
    // B operand: own registers (image of group c').  Accumulator = -(tile) in the standard orientation.
    double* A11 = A10 + 128;
    auto tile = [&](int c, int cp, const gs_d4 (&Pq)[8]) {
        gs_d4 acc;
#pragma unroll
        for (int x = 0; x < 4; ++x) acc[x] = -A11[(int64_t)(16 * c + fq + 4 * x) * ld + 16 * cp + fr];
        gs_d4 buf[8];
        gs_sib_fetch(buf, Lsib, c, lane);
        gs_sib_apply(acc, buf, Pq);
#pragma unroll
        for (int x = 0; x < 4; ++x) A11[(int64_t)(16 * c + fq + 4 * x) * ld + 16 * cp + fr] = -acc[x];
    };
    for (int c = g0; c < 8; ++c) tile(c, g0, Pa);
    for (int c = g1; c < 8; ++c) tile(c, g1, Pb);
    __threadfence();
    __syncthreads();
    bad = gs_diag_block<false>(A11, ld, (double*)nullptr, Ltab + GS_LTAB, logdet + 1, diag0 + 128, nullptr, wsd);
    if (bad && t == 0) *info = col0 + 128 + bad;
}

__global__ __launch_bounds__(256, 2) void k_potrf_diag256(double* A, int64_t ld, double* Ltab, double* Lsib, double* logdet, int* info,
                                                       int col0, const double* diag0, unsigned long long* stamps) {
    __shared__ __attribute__((aligned(16))) double wsd[GS_D2C_WS];
    gs_potrf_diag256_body(A, ld, Ltab, Lsib, logdet, info, col0, diag0, stamps, wsd);
}

// ---- grouped chain kernels (see k_gemm_ld3g): outer step `step` of workspace `q` per entry, all workspaces of a group at fixed
// strides from the first.  One workgroup per entry (diagonal super-block); one wave per 16 rows below it of every entry (panels).
#define GS_WVC_MAX 24
struct gs_wv_pool {
    double* A; int64_t strideA, ld;          // augmented matrices, (np + 16) x ld each
    double* Ltab; double* Lsib;              // T x GS_LTAB, (T / 2 + 1) x GS_LSIB per workspace
    double* logdet; double* diag0;           // T, np per workspace
    int* info;                               // 1 per workspace
    double* res;                             // 258 per workspace (k_finalize_g)
    int np, T;
};
struct gs_wv_chain_args {
    gs_wv_pool p;
    int n, pad;
    short q[GS_WVC_MAX], step[GS_WVC_MAX];
    int end[GS_WVC_MAX];                     // k_panel256g: running counts of 16-row groups
};
__global__ __launch_bounds__(256, 2) void k_potrf_diag256g(const gs_wv_chain_args a) {
    __shared__ __attribute__((aligned(16))) double wsd[GS_D2C_WS];
    const int e = (int)blockIdx.x;
    const int64_t q = a.q[e];
    const int b = 2 * a.step[e];
    const int64_t c = (int64_t)b * GS_NB;
    gs_potrf_diag256_body(a.p.A + q * a.p.strideA + c * a.p.ld + c, a.p.ld, a.p.Ltab + (q * a.p.T + b) * GS_LTAB,
                          a.p.Lsib + (q * (a.p.T / 2 + 1) + b / 2) * GS_LSIB, a.p.logdet + q * a.p.T + b, a.p.info + q, (int)c,
                          a.p.diag0 + q * a.p.np + c, (unsigned long long*)nullptr, wsd);
}
__global__ __launch_bounds__(64, 2) void k_panel256g(const gs_wv_chain_args a) {
    const int bid = (int)blockIdx.x;
    int e = 0;
    while (e + 1 < a.n && bid >= a.end[e]) ++e;
    const int first = e ? a.end[e - 1] : 0;
    const int64_t q = a.q[e];
    const int b = 2 * a.step[e];
    const int64_t c0 = (int64_t)b * GS_NB, r2 = c0 + 2 * GS_NB;
    const int M = a.p.np + GS_BORDER - (int)r2;
    gs_panel256_body(a.p.A + q * a.p.strideA + r2 * a.p.ld + c0, a.p.ld, M, a.p.Ltab + (q * a.p.T + b) * GS_LTAB,
                     a.p.Lsib + (q * (a.p.T / 2 + 1) + b / 2) * GS_LSIB, a.p.Ltab + (q * a.p.T + b + 1) * GS_LTAB,
                     (unsigned long long*)nullptr, (unsigned long long*)nullptr, bid - first);
}
// The same with W = 4 (or 8) waves per workgroup, each on its own 16-row group.  Single-wave workgroups are spread round-robin
// over the CUs, and one 224-register panel wave on a SIMD is enough to keep a whole bulk workgroup (2 waves on EACH of the CU's 4
// SIMDs) off that CU: a thin spread of panel waves costs the trailing updates of the other groups up to a third of every CU it
// touches.  Four waves per workgroup land on ONE CU and use the evicted workgroup's room on all four SIMDs.
template <int W, bool TR = false>
__global__ __launch_bounds__(64 * W, W == 4 ? 2 : 1) void k_panel256gw(const gs_wv_chain_args a) {
    __shared__ __attribute__((aligned(16))) double tiles[TR ? W * GS_PT_TILE : 2];
    const int grp = (int)blockIdx.x * W + (int)(threadIdx.x >> 6);
    int e = 0;
    while (e + 1 < a.n && grp >= a.end[e]) ++e;
    if (grp >= a.end[a.n - 1]) return;
    const int first = e ? a.end[e - 1] : 0;
    const int64_t q = a.q[e];
    const int b = 2 * a.step[e];
    const int64_t c0 = (int64_t)b * GS_NB, r2 = c0 + 2 * GS_NB;
    const int M = a.p.np + GS_BORDER - (int)r2;
    gs_panel256_body(a.p.A + q * a.p.strideA + r2 * a.p.ld + c0, a.p.ld, M, a.p.Ltab + (q * a.p.T + b) * GS_LTAB,
                     a.p.Lsib + (q * (a.p.T / 2 + 1) + b / 2) * GS_LSIB, a.p.Ltab + (q * a.p.T + b + 1) * GS_LTAB,
                     (unsigned long long*)nullptr, (unsigned long long*)nullptr, grp - first,
                     TR ? tiles + (threadIdx.x >> 6) * GS_PT_TILE : (double*)nullptr);
}
// entering evaluations: border rows <- RHS^T (k_set_border), grid ((np + 16) / 256 rounded up, entries)
struct gs_wv_zsets { int64_t off[GS_WVC_MAX]; };         // per entry: offset (doubles) of its right-hand-side set in Z
__global__ __launch_bounds__(256) void k_set_border_g(const gs_wv_chain_args a, int n, const double* Z, int k, const gs_wv_zsets zs) {
    const int64_t q = a.q[blockIdx.y];
    Z += zs.off[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.p.np + GS_BORDER) return;
    double* A = a.p.A + q * a.p.strideA;
#pragma unroll
    for (int c = 0; c < GS_BORDER; ++c)
        A[(int64_t)(a.p.np + c) * a.p.ld + i] = (c < k && i < n) ? Z[(int64_t)i * k + c] : 0.0;
}
// entering evaluations: diag0 <- the diagonal before the factorisation touches it, info <- 0 (grid: (np / 256 rounded up, entries))
__global__ __launch_bounds__(256) void k_wave_begin(const gs_wv_chain_args a) {
    const int64_t q = a.q[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.p.np) a.p.diag0[q * a.p.np + i] = a.p.A[q * a.p.strideA + (int64_t)i * a.p.ld + i];
    if (i == 0) a.p.info[q] = 0;
}

// ------------------------------------------------------------------------------------------------
// PERSISTENT CHAIN (round 3): the dependent chain of ONE factorisation as one resident kernel on CUs of its own.
//
// One factorisation alone is bound by the chain diag -> panel -> sibling update -> diag -> panel -> look-ahead update of every
// 256-column outer step, not by the bulk update: as host-enqueued launches every link queued behind bulk
// workgroups for a CU slot, shared its SIMDs' matrix pipes with them (k_potrf_diag 43-96 us beside the bulk update, 31 alone)
// and ran over ALL rows below the panel although the next diagonal blocks only need the rows just below it.  Here the chain
// is cut down to a WINDOW of W rows under the panel and runs as ONE kernel of 1 + W / 64 workgroups that each hold a CU
// alone (128 KB of LDS: no bulk workgroup fits beside them) and talk through flags in global memory:
//   workgroup 0 (D role)       per outer step s (block columns k = 2 s, k + 1):  D(k) -> T0 | rows of block k + 1 solved
//                              against it (tables in LDS) -> TL | A11 -= L10 L10^T | D(k + 1) -> T1
//   workgroups 1.. (P role)    one wave per 16-row group of the window [r2, r2 + W): X_k = B_k L_kk^-T (after T0), sibling
//                              update (after TL), X_k+1 (after T1), rows + operand images published; then the window's share of
//                              the trailing update, C[window rows, next panel's columns] -= P P^T (K = 256), as 32 x 32 tasks
//                              over the published images, the next diagonal block's tasks first (counters UD0 / UD1 / UR)
// Everything M-proportional -- the panel of the rows below the window (k_panel256, gated on T1), the update of the next
// panel's columns below the window (A), of the panel after it (B) and of the far region (Far) -- stays host-enqueued on two
// streams and meets the chain through the same flags: a one-wave k_wait_flag in front of a launch holds its stream until the chain has set the flag, one-thread
// k_signal launches tell the chain that A(s) / B(s) have finished.  The regions are a partition of the trailing update of
// the host-enqueued schedule and every element receives the same products in the same ascending order: results are
// bit-identical to it (tests/test_gpu_parity.py).
//
// Hand-off discipline (MI355X_MICROARCH.md, inter-workgroup visibility): published bytes are stored write-through (relaxed
// agent-scope atomic stores = global_store ... sc1), every storing wave drains vmcnt, (workgroup barrier,) ONE lane stores the
// flag / adds to the counter; a consumer polls relaxed, then ONE agent-scope acquire, s_waitcnt vmcnt(0), (barrier,) plain loads.
// Every spin is bounded (GS_CH_TIMEOUT): on expiry flags[GS_FL_ABORT] = 1 and every party leaves at its next wait.
// ------------------------------------------------------------------------------------------------
// a wave-uniform pointer made opaque to the optimiser, in SGPRs: inside the persistent loops LICM otherwise hoists hundreds of
// per-lane 64-bit table addresses (base + lane + constant) out of the loop and spills them (1000 spilled VGPRs measured)
template <class T>
__device__ __forceinline__ T* gs_uniform_ptr(T* p) {
    const unsigned long long v = (unsigned long long)p;          // (readfirstlane: uniform by construction, whatever the
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);               // divergence analysis thinks)
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    p = (T*)(((unsigned long long)hi << 32) | lo);
    asm volatile("" : "+s"(p));
    return p;
}

struct gs_chain_args {
    double* A; int64_t ld; int np, naug, S, W;
    double* Ltab; double* Lsib; double* logdet; const double* diag0; int* info;
    double* dump;                  // [2][GS_CH_GMAX][16][4][64]: operand images of the window's solved rows, by step parity
    unsigned* flags;               // gs_fl_count(S) words, zeroed before the launch
    const unsigned* fbwant;        // S words: how many first-256-column tiles the host-enqueued trailing update of step s counts in FB[s]
    int test_abort;                // test hook (option "chain_test_abort"): the D role gives up at this outer step as if a wait had timed out
    unsigned long long* stamps;    // S x GS_CH_STAMPS realtime stamps, or NULL
};

// window geometry of outer step s: Gs row groups [r2, r2 + 16 Gs), Gc column groups of the next panel
__device__ __forceinline__ void gs_ch_geom(int np, int naug, int W, int s, int& r2, int& Gs, int& Gc) {
    r2 = 256 * (s + 1);
    const int wend = min(r2 + W, naug);
    Gs = (wend - r2) / 16;
    Gc = min(16, (naug - r2) / 16);
}
// tiles of the first 256 columns of outer step s's host-enqueued trailing update (k_gemm_ld3, nfirst): what FB[s] counts up to
__host__ __device__ inline unsigned gs_ch_nfirst(int naug, int s) {
    const int m3 = naug - 256 * (s + 2);
    return m3 > 0 ? (unsigned)(4 * ((m3 + 127) / 128) - 2) : 0u;
}
// number of 32 x 32 update tasks (I, J), J <= I, with Ilo <= I < Ihi, in a window of Gs row and Gc column groups
__device__ __forceinline__ int gs_ch_ntasks(int Gs, int Gc, int Ilo, int Ihi) {
    const int NI = (Gs + 1) / 2, NJ = (Gc + 1) / 2;
    int c = 0;
    for (int I = Ilo; I < min(Ihi, NI); ++I) c += min(I + 1, NJ);
    return c;
}

__device__ __forceinline__ void gs_panel16_store_wt(const gs_d4 (&P)[8], double* rows, int64_t ld, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int x = 0; x < 4; ++x) gs_st_wt(rows + (int64_t)fr * ld + 16 * k + fq + 4 * x, -P[k][x]);
}
__device__ __forceinline__ void gs_image_store_wt(const gs_d4 (&P)[8], double* img, int lane) {      // img: [kb][x][lane]
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int x = 0; x < 4; ++x) gs_st_wt(img + (k * 4 + x) * 64 + lane, P[k][x]);
}

// ---- D role: the 256 x 256 diagonal super-block of every outer step (k_potrf_diag256's arithmetic, tables kept in LDS)
__device__ __forceinline__ void gs_chain_diag_role(const gs_chain_args& a, double* wsd, volatile int* sh) {
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int S = a.S;
    unsigned* fl = a.flags;
    for (int s = 0; s < S; ++s) {
        const int k = 2 * s;
        const int64_t c0 = 256 * (int64_t)s, ld = a.ld;
        double* A00 = gs_uniform_ptr(a.A + c0 * ld + c0);
        double* A10 = gs_uniform_ptr(A00 + 128 * ld);
        double* A11 = gs_uniform_ptr(A10 + 128);
        double* tab0 = gs_uniform_ptr(a.Ltab + (size_t)k * GS_LTAB);
        double* tab1 = gs_uniform_ptr(tab0 + GS_LTAB);
        double* sib = gs_uniform_ptr(a.Lsib + (size_t)s * GS_LSIB);
        unsigned long long* st = a.stamps ? a.stamps + (size_t)s * GS_CH_STAMPS : nullptr;
        int pr2 = 0, pGs = 0, pGc = 0;
        if (s > 0) gs_ch_geom(a.np, a.naug, a.W, s - 1, pr2, pGs, pGc);
        if (st && t == 0) st[0] = __builtin_amdgcn_s_memrealtime();
        if (a.test_abort > 0 && s == a.test_abort) {          // (tests only: exercise the give-up path of every party and of the host)
            if (t == 0) gs_flag_st(fl + GS_FL_ABORT, 1u);
            return;
        }
        if (s > 0 && !gs_wg_wait_ge(fl + gs_fl(GS_FL_UD0, S, s - 1), (unsigned)gs_ch_ntasks(pGs, pGc, 0, 4), fl, sh)) return;
        if (st && t == 0) st[1] = __builtin_amdgcn_s_memrealtime();
        int bad = gs_diag_block<true>(A00, ld, (double*)nullptr, (double*)nullptr, a.logdet + k, a.diag0 + c0, nullptr, wsd);
        if (bad) {                                  // uniform
            if (t == 0) {
                *a.info = (int)c0 + bad;
                gs_flag_st(fl + GS_FL_ABORT, 2u);
            }
            return;
        }
        // tables of block k to global memory, write-through; their flag goes out below, behind the row solves (the stores drain
        // meanwhile: the P waves have the ~70 us until T1 for their first solve and sibling update)
        for (int i = t; i < GS_LTAB; i += 256) gs_st_wt(tab0 + i, wsd[i]);
        // ---- L10: row groups w and 7 - w of block row k + 1 against the tables in LDS
        if (s > 0 && !gs_wg_wait_ge(fl + gs_fl(GS_FL_UD1, S, s - 1), (unsigned)gs_ch_ntasks(pGs, pGc, 4, 8), fl, sh)) return;
        if (st && t == 0) st[3] = __builtin_amdgcn_s_memrealtime();
        const int g0 = w, g1 = 7 - w;
        gs_d4 Pa[8], Pb[8];
        gs_panel16_load(Pa, A10 + (int64_t)(16 * g0) * ld, ld, 16, lane);
        gs_panel16_load(Pb, A10 + (int64_t)(16 * g1) * ld, ld, 16, lane);
        gs_panel16_solve2(Pa, Pb, wsd, lane);
        gs_drain();
        __syncthreads();                            // tables published; every wave is through with them: the LDS takes the images of L10
        if (t == 0) gs_flag_st(fl + gs_fl(GS_FL_T0, S, s), 1u);
        if (st && t == 0) st[2] = __builtin_amdgcn_s_memrealtime();
        // (L10's rows go back to the matrix from the published image, by P waves 0..7: gs_chain_panel_role)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                wsd[((g0 * 8 + kb) * 4 + x) * 64 + lane] = Pa[kb][x];
                wsd[((g1 * 8 + kb) * 4 + x) * 64 + lane] = Pb[kb][x];
                gs_st_wt(sib + ((g0 * 8 + kb) * 4 + x) * 64 + lane, Pa[kb][x]);
                gs_st_wt(sib + ((g1 * 8 + kb) * 4 + x) * 64 + lane, Pb[kb][x]);
            }
        __syncthreads();                            // images in LDS (the copies for the other workgroups drain behind the update)
        if (st && t == 0) st[4] = __builtin_amdgcn_s_memrealtime();
        // ---- A11 -= L10 L10^T: micro-tiles (c, c') for c' in {g0, g1}, c >= c' (k_gemm_nt's arithmetic: -C + sum, ascending k)
        auto tile = [&](int c, int cp, const gs_d4 (&Pq)[8]) {
            gs_d4 acc;
#pragma unroll
            for (int x = 0; x < 4; ++x) acc[x] = -A11[(int64_t)(16 * c + fq + 4 * x) * ld + 16 * cp + fr];
#pragma unroll
            for (int kb = 0; kb < 8; ++kb)
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wsd[((c * 8 + kb) * 4 + x) * 64 + lane], Pq[kb][x], acc, 0, 0, 0);
#pragma unroll
            for (int x = 0; x < 4; ++x) A11[(int64_t)(16 * c + fq + 4 * x) * ld + 16 * cp + fr] = -acc[x];
        };
        for (int c = g0; c < 8; ++c) tile(c, g0, Pa);
        for (int c = g1; c < 8; ++c) tile(c, g1, Pb);
        gs_drain();
        __syncthreads();
        if (t == 0) gs_flag_st(fl + gs_fl(GS_FL_TL, S, s), 1u);
        if (st && t == 0) st[5] = __builtin_amdgcn_s_memrealtime();
        bad = gs_diag_block<true>(A11, ld, (double*)nullptr, (double*)nullptr, a.logdet + k + 1, a.diag0 + c0 + 128, nullptr, wsd);
        if (bad) {
            if (t == 0) {
                *a.info = (int)c0 + 128 + bad;
                gs_flag_st(fl + GS_FL_ABORT, 2u);
            }
            return;
        }
        for (int i = t; i < GS_LTAB; i += 256) gs_st_wt(tab1 + i, wsd[i]);
        gs_drain();
        __syncthreads();
        if (t == 0) gs_flag_st(fl + gs_fl(GS_FL_T1, S, s), 1u);
        if (st && t == 0) st[6] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---- P role: one wave = one 16-row group of the window per outer step, then its share of the window's update tasks.
// A 32 x 32 update task (I, J): C[rows of groups 2I, 2I+1][columns of groups 2J, 2J+1] -= P P^T over the panel's 256 columns
// (16 k-blocks of operand images); held in four accumulators that start as -C (the bulk tiles' arithmetic).
struct gs_utask {
    int I, J, gi0, gj0;
    bool va1, vb1, v01, v10, v11;
    double* C0;
    const double *dA0, *dA1, *dB0, *dB1;
    gs_d4 c00, c01, c10, c11;
};

__device__ __forceinline__ bool gs_utask_decode(gs_utask& u, int tk, int Gs, int Gc, double* A, int64_t ld, int r2, const double* dump,
                                                int lane) {
    const int NI = (Gs + 1) / 2, NJ = (Gc + 1) / 2;
    int I = 0, J = -1, seen = 0;
    for (I = 0; I < NI; ++I) {
        const int c = min(I + 1, NJ);
        if (tk < seen + c) { J = tk - seen; break; }
        seen += c;
    }
    if (J < 0) return false;
    u.I = I; u.J = J; u.gi0 = 2 * I; u.gj0 = 2 * J;
    u.va1 = u.gi0 + 1 < Gs;
    u.vb1 = u.gj0 + 1 < Gc;
    // micro-tile (a, b): rows of group gi0 + a, columns of group gj0 + b; on a diagonal task only the lower ones
    u.v01 = u.vb1 && u.gi0 >= u.gj0 + 1;
    u.v10 = u.va1;
    u.v11 = u.va1 && u.vb1;
    u.C0 = gs_uniform_ptr(A + (int64_t)(r2 + 16 * u.gi0) * ld + r2 + 16 * u.gj0);
    u.dA0 = gs_uniform_ptr(dump + (size_t)u.gi0 * 16 * 256) + lane;
    u.dA1 = gs_uniform_ptr(dump + (size_t)(u.va1 ? u.gi0 + 1 : u.gi0) * 16 * 256) + lane;
    u.dB0 = gs_uniform_ptr(dump + (size_t)u.gj0 * 16 * 256) + lane;
    u.dB1 = gs_uniform_ptr(dump + (size_t)(u.vb1 ? u.gj0 + 1 : u.gj0) * 16 * 256) + lane;
    return true;
}

// the flags of the four row groups a task multiplies have reached `want` (1: first 128 panel columns published, 2: all 256)
__device__ __forceinline__ bool gs_utask_poll(const gs_utask& u, unsigned* fl, int S, int s, unsigned want) {
    if (!gs_poll_ge(fl + gs_fl_wg(S, s, u.gi0), want, fl)) return false;
    if (u.va1 && !gs_poll_ge(fl + gs_fl_wg(S, s, u.gi0 + 1), want, fl)) return false;
    if (!gs_poll_ge(fl + gs_fl_wg(S, s, u.gj0), want, fl)) return false;
    if (u.vb1 && !gs_poll_ge(fl + gs_fl_wg(S, s, u.gj0 + 1), want, fl)) return false;
    return true;
}

__device__ __forceinline__ void gs_utask_load(gs_utask& u, int64_t ld, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int64_t ro = (int64_t)(fq + 4 * x) * ld + fr;
        u.c00[x] = -u.C0[ro];
        u.c01[x] = u.v01 ? -u.C0[ro + 16] : 0.0;
        u.c10[x] = u.v10 ? -u.C0[ro + 16 * ld] : 0.0;
        u.c11[x] = u.v11 ? -u.C0[ro + 16 * ld + 16] : 0.0;
    }
}

// k-blocks [kb0, kb1) in ascending order, operands one k-block ahead
__device__ __forceinline__ void gs_utask_accumulate(gs_utask& u, int kb0, int kb1) {
    gs_d4 a0, a1, b0, b1, na0, na1, nb0, nb1;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        a0[x] = u.dA0[(kb0 * 4 + x) * 64];
        a1[x] = u.dA1[(kb0 * 4 + x) * 64];
        b0[x] = u.dB0[(kb0 * 4 + x) * 64];
        b1[x] = u.dB1[(kb0 * 4 + x) * 64];
    }
#pragma unroll 1
    for (int kb = kb0; kb < kb1; ++kb) {
        const int kn = kb + 1 < kb1 ? kb + 1 : kb;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            na0[x] = u.dA0[(kn * 4 + x) * 64];
            na1[x] = u.dA1[(kn * 4 + x) * 64];
            nb0[x] = u.dB0[(kn * 4 + x) * 64];
            nb1[x] = u.dB1[(kn * 4 + x) * 64];
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            u.c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[x], b0[x], u.c00, 0, 0, 0);
            if (u.v01) u.c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[x], b1[x], u.c01, 0, 0, 0);
            if (u.v10) u.c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[x], b0[x], u.c10, 0, 0, 0);
            if (u.v11) u.c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[x], b1[x], u.c11, 0, 0, 0);
        }
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
}

__device__ __forceinline__ void gs_utask_store(const gs_utask& u, int64_t ld, unsigned* fl, int S, int s, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int64_t ro = (int64_t)(fq + 4 * x) * ld + fr;
        gs_st_wt(u.C0 + ro, -u.c00[x]);
        if (u.v01) gs_st_wt(u.C0 + ro + 16, -u.c01[x]);
        if (u.v10) gs_st_wt(u.C0 + ro + 16 * ld, -u.c10[x]);
        if (u.v11) gs_st_wt(u.C0 + ro + 16 * ld + 16, -u.c11[x]);
    }
    gs_drain();
    if (lane == 0) gs_flag_add(fl + gs_fl(u.I < 4 ? GS_FL_UD0 : (u.I < 8 ? GS_FL_UD1 : GS_FL_UR), S, s));
}

// Per outer step, wave pw (its workgroup's four waves meet at two barriers: the second block's tables are staged in LDS once):
//   rows ready -> [T0] X_k, published (group flag = 1) -> [TL] sibling update; L10's rows to the matrix (waves 0..7) ->
//   FIRST HALF of its first update task (k-blocks 0..7 need only the X_k images; accumulators stay in registers) ->
//   [T1] tables of block k + 1 into LDS (the four waves a quarter each) -> X_k+1, published (group flag = 2) ->
//   second half of that task -> its other tasks in full.
// What is left on the chain's critical path between T1 and the next diagonal block: one solve from LDS, one publish, 128 MFMAs.
__device__ __forceinline__ void gs_chain_panel_role(const gs_chain_args& a, int pw, int NPW, int lane, double* tabl) {
    const int fr = lane & 15, fq = lane >> 4;
    const int wq = pw & 3;                       // wave within its workgroup
    const int S = a.S;
    const int64_t ld = a.ld;
    unsigned* fl = a.flags;
    for (int s = 0; s < S; ++s) {
        int r2, Gs, Gc;
        gs_ch_geom(a.np, a.naug, a.W, s, r2, Gs, Gc);
        const int64_t c0 = 256 * (int64_t)s;
        const double* tab0 = gs_uniform_ptr(a.Ltab + (size_t)(2 * s) * GS_LTAB);
        const double* tab1 = gs_uniform_ptr(a.Ltab + (size_t)(2 * s + 1) * GS_LTAB);
        const double* sib = gs_uniform_ptr(a.Lsib + (size_t)s * GS_LSIB);
        double* dump = a.dump + (size_t)(s & 1) * GS_CH_GMAX * 16 * 256;
        unsigned long long* st = (a.stamps && pw == 0) ? a.stamps + (size_t)s * GS_CH_STAMPS : nullptr;
        int pr2 = 0, pGs = 0, pGc = 0;
        if (s > 0) gs_ch_geom(a.np, a.naug, a.W, s - 1, pr2, pGs, pGc);
        const int g = pw;
        const bool has = g < Gs;                 // NPW = W / 16 >= Gs: a wave owns at most one row group
        double* rows = gs_uniform_ptr(a.A + (int64_t)(r2 + 16 * (has ? g : 0)) * ld + c0);
        double* img = gs_uniform_ptr(dump + (size_t)(has ? g : 0) * 16 * 256);
        gs_d4 P0[8], P1[8];
        // ---- X_k = B_k L_kk^-T  (k_panel256's arithmetic throughout)
        if (has) {
            if (s > 0) {
                // these rows' entries in panel s's columns: last updated by the window tasks of step s - 1 (rows that were
                // in that window: its groups 16 ..) or by the host-enqueued update A(s - 1) (rows below it)
                const bool in_prev = g + 16 < pGs;
                const unsigned* f = in_prev ? fl + gs_fl(GS_FL_UR, S, s - 1) : fl + gs_fl(GS_FL_FA, S, s - 1);
                const unsigned want = in_prev ? (unsigned)gs_ch_ntasks(pGs, pGc, 8, 1 << 20) : 1u;
                if (!gs_wait_ge(f, want, fl)) return;
            }
            if (st) st[8] = __builtin_amdgcn_s_memrealtime();
            gs_panel16_load(P0, rows, ld, 16, lane);
            if (!gs_wait_ge(fl + gs_fl(GS_FL_T0, S, s), 1u, fl)) return;
            if (st) st[9] = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_sched_barrier(0);
            gs_panel16_solve_g(P0, tab0, lane);
            gs_panel16_store_wt(P0, rows, ld, lane);
            gs_image_store_wt(P0, img, lane);
            gs_drain();
            if (lane == 0) gs_flag_st(fl + gs_fl_wg(S, s, g), 1u);
            __builtin_amdgcn_sched_barrier(0);          // the second 128 columns are fetched only now (as k_panel256)
            gs_panel16_load(P1, rows + 128, ld, 16, lane);
        }
        if (has || pw < 8) {
            if (!gs_wait_ge(fl + gs_fl(GS_FL_TL, S, s), 1u, fl)) return;
        }
        if (has) {
            gs_sib_update_lean(P1, P0, sib, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (pw < 8) {
            // L(k+1, k) back into the matrix, from its operand image (the diagonal workgroup only publishes the image: 128
            // scattered stores per lane off its critical path); nobody reads these rows before the factorisation ends
            double* l10 = gs_uniform_ptr(a.A + (c0 + 128 + 16 * pw) * ld + c0);
            const double* im = gs_uniform_ptr(sib + (size_t)pw * 8 * 256) + lane;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb)
#pragma unroll
                for (int x = 0; x < 4; ++x) l10[(int64_t)fr * ld + 16 * kb + fq + 4 * x] = -im[(kb * 4 + x) * 64];
        }
        if (st) st[10] = __builtin_amdgcn_s_memrealtime();
        // ---- first half of this wave's first update task, while the diagonal workgroup factors block k + 1
        const int ntask = gs_ch_ntasks(Gs, Gc, 0, 1 << 20);
        gs_utask u;
        bool early = false;
        const bool have_task = pw < ntask && gs_utask_decode(u, pw, Gs, Gc, a.A, ld, r2, dump, lane);
        const unsigned fb_want = s > 0 ? a.fbwant[s - 1] : 0u;     // B(s - 1): C's last host-enqueued update, counted per tile
        if (have_task && gs_flag_ld(fl + gs_fl(GS_FL_FB, S, s > 0 ? s - 1 : 0)) >= fb_want) {      // (C is up to date already: else later, in full)
            if (!gs_utask_poll(u, fl, S, s, 1u)) return;
            gs_acquire();
            gs_utask_load(u, ld, lane);
            gs_utask_accumulate(u, 0, 8);
            early = true;
        }
        // ---- tables of block k + 1 into LDS, X_k+1 = B_k+1 L_k+1,k+1^-T
        if (!gs_wait_ge(fl + gs_fl(GS_FL_T1, S, s), 1u, fl)) return;
        if (st) st[11] = __builtin_amdgcn_s_memrealtime();
        __syncthreads();                             // the previous step's readers of the LDS tables are through
        gs_load_ltab_direct(tabl, tab1, wq, lane);
        gs_drain();
        __syncthreads();
        if (has) {
            gs_panel16_solve(P1, tabl, lane);
            gs_panel16_store_wt(P1, rows + 128, ld, lane);
            gs_image_store_wt(P1, img + 8 * 256, lane);
            gs_drain();
            if (lane == 0) {
                gs_flag_st(fl + gs_fl_wg(S, s, g), 2u);
                if (g < 16) gs_flag_add(fl + gs_fl(GS_FL_WTOP, S, s));
                gs_flag_add(fl + gs_fl(GS_FL_WALL, S, s));
            }
        }
        if (st) st[12] = __builtin_amdgcn_s_memrealtime();
        // ---- the window's share of the trailing update, tasks in ascending I (the next diagonal block's first)
        for (int tk = pw; tk < ntask; tk += NPW) {
            const bool first = tk == pw;
            if (!first && !gs_utask_decode(u, tk, Gs, Gc, a.A, ld, r2, dump, lane)) break;
            if (!gs_utask_poll(u, fl, S, s, 2u)) return;
            if (s > 0 && !gs_poll_ge(fl + gs_fl(GS_FL_FB, S, s - 1), fb_want, fl)) return;
            gs_acquire();
            if (st && first) st[13] = __builtin_amdgcn_s_memrealtime();
            if (first && early) {
                gs_utask_accumulate(u, 8, 16);
            } else {
                gs_utask_load(u, ld, lane);
                gs_utask_accumulate(u, 0, 16);
            }
            gs_utask_store(u, ld, fl, S, s, lane);
            if (st && first) st[14] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

__global__ __launch_bounds__(256, 1) void k_chain(gs_chain_args a) {
    extern __shared__ __attribute__((aligned(16))) double wsd[];
    __shared__ int sh_ok;
    __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x == 0) gs_flag_add(a.flags + GS_FL_RESIDENT);
    if (blockIdx.x == 0) {
        gs_chain_diag_role(a, wsd, &sh_ok);
    } else {
        const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        gs_chain_panel_role(a, ((int)blockIdx.x - 1) * 4 + w, ((int)gridDim.x - 1) * 4, threadIdx.x & 63, wsd);
    }
}

