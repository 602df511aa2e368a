// Stand-alone microbenchmark (round 5): what separates the bulk tile's K loop (0.86-0.90 of the fp64 MFMA peak in steady state) from a loop of
// MFMAs alone (0.99)?  One workgroup = 8 waves with the tile's 32 x 32 accumulator block each; variants add the loop's other ingredients one by one.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/probe_mfma_lds tools/probe_mfma_lds.hip && /tmp/probe_mfma_lds
// Not part of the product; results in profiles/r05_mfma_lds_probe.log.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// V: 0 MFMAs only (operands in registers); 1 + fragment reads from LDS (two ds_read2st64_b64 per k-step, as the tile); 2 + a barrier per chunk of
// 16 MFMAs; 3 + three LDS-direct loads per wave and chunk (L2-resident source) with the tile's vmcnt(0) in front of the barrier;
// 4 = 1 with ALL of a chunk's fragments read first (8 reads, then 16 MFMAs); 5 = 1 with the reads of step s + 1 issued BETWEEN the MFMAs of step s;
// 6 = 1 with ds_read_b128-style pairs (half the LDS instructions for the same bytes, k-pairs adjacent)
// V = 7: variant 3 with the tile's REAL operand addressing: workgroup b multiplies tile (bm, bn) of a big row-major matrix pair (lda doubles per row), every
// wave loading its three 8-row slices of the current 16-column chunk (8 lanes x 16 B per row), chunk after chunk along K; BIG: how many chunks before
// the walk wraps (128 = a K = 2048 tile from HBM / Infinity Cache, 1 = every chunk the same 24 KB: L2-hot)
template <int WRAP, int MODE = 0>
__global__ __launch_bounds__(512) void k_probe_mem(double* out, const double* A, int64_t lda, int tiles_m, int chunks, int idle_mask = 0) {
    extern __shared__ double lds[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int i = t; i < 6144; i += 512) lds[i] = 1.0 + 1e-6 * (i % 977);
    __syncthreads();
    d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
    const int wm = w & 3, wn = w >> 2;
    const double* sA = lds + wm * 512 + lane;
    const double* sB = lds + 2048 + wn * 512 + lane;
    const int bm = blockIdx.x % tiles_m, bn = (blockIdx.x / tiles_m) % (2 * tiles_m);
    const int lrow = lane >> 3, lg = lane & 7;
    // A-side rows 128 bm + 16 w + {0..7, 8..15}, B-side rows 64 bn + 8 w + lrow (taken from the same matrix, another row range)
    const double* g0 = A + (int64_t)(128 * bm + 16 * w + lrow) * lda + 2 * lg;
    const double* g1 = g0 + 8 * lda;
    const double* g2 = A + (int64_t)(64 * bn + 8 * w + lrow) * lda + 2 * lg;
    // (MODE & 1: `idle` is false for every wave, but only the hardware knows)
    const bool idle = (MODE & 1) && __builtin_amdgcn_readfirstlane((idle_mask >> w) & 1);
    auto mm = [&](const double (&af)[2], const double (&bf)[2]) {
        if (idle) return;
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0], bf[0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0], bf[1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[1], bf[0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[1], bf[1], acc[1][1], 0, 0, 0);
    };
    auto loads = [&](int c) {
        double* dst = lds + (c & 1) * 3072 + w * 384;
        const int kc = 16 * ((c + 2) % WRAP);
        __builtin_amdgcn_global_load_lds(g0 + kc, dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(g1 + kc, dst + 128, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(g2 + kc, dst + 256, 16, 0, 0);
    };
    if constexpr ((MODE & 4) != 0) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 r0 = {0.0, 0.0}, r1 = r0, r2 = r0;
        for (int c = 0; c < chunks; ++c) {
            const int st = (c & 1) * 3072;
            double af[2][2], bf[2][2];
            af[0][0] = sA[st]; af[0][1] = sA[st + 64]; bf[0][0] = sB[st]; bf[0][1] = sB[st + 64];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1, kn = 128 * ((ks + 1) & 3);
                __builtin_amdgcn_sched_barrier(0);
                af[nxt][0] = sA[st + kn]; af[nxt][1] = sA[st + 64 + kn]; bf[nxt][0] = sB[st + kn]; bf[nxt][1] = sB[st + 64 + kn];
                __builtin_amdgcn_sched_barrier(0);
                mm(af[cur], bf[cur]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // the registers hold chunk c + 1's operands (requested a chunk ago): into the stage chunk c - 1 was read from ... wait: the OTHER stage is
            // being read by nobody only behind the barrier, so write behind it and pay a second barrier (the classic two-barrier register pipeline)
            __syncthreads();
            d2* dst = reinterpret_cast<d2*>(lds + (c & 1) * 3072 + w * 384) + lane;
            dst[0] = r0; dst[64] = r1; dst[128] = r2;
            const int kc = 16 * ((c + 2) % WRAP);
            r0 = *reinterpret_cast<const d2*>(g0 + kc);
            r1 = *reinterpret_cast<const d2*>(g1 + kc);
            r2 = *reinterpret_cast<const d2*>(g2 + kc);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (r0[0] + r1[0] + r2[0] == 1.2345e301) out[t] = r0[1];
    } else if constexpr ((MODE & 2) == 0) {
        for (int c = 0; c < chunks; ++c) {
            const int st = (c & 1) * 3072;
            double af[2][2], bf[2][2];
            af[0][0] = sA[st]; af[0][1] = sA[st + 64]; bf[0][0] = sB[st]; bf[0][1] = sB[st + 64];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1, kn = 128 * ((ks + 1) & 3);
                __builtin_amdgcn_sched_barrier(0);
                af[nxt][0] = sA[st + kn]; af[nxt][1] = sA[st + 64 + kn]; bf[nxt][0] = sB[st + kn]; bf[nxt][1] = sB[st + 64 + kn];
                __builtin_amdgcn_sched_barrier(0);
                mm(af[cur], bf[cur]);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            loads(c);
        }
    } else {
        // the tile's order: fragments one k-step ahead in two register sets across the chunk boundary
        double afA[2], bfA[2], afB[2], bfB[2];
        auto ldf = [&](double (&af)[2], double (&bf)[2], int st, int ks) {
            af[0] = sA[st + 128 * ks]; af[1] = sA[st + 128 * ks + 64]; bf[0] = sB[st + 128 * ks]; bf[1] = sB[st + 128 * ks + 64];
        };
        ldf(afA, bfA, 0, 0);
        for (int c = 0; c < chunks; ++c) {
            const int st = (c & 1) * 3072;
            __builtin_amdgcn_sched_barrier(0);
            ldf(afB, bfB, st, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(afA, bfA);
            __builtin_amdgcn_sched_barrier(0);
            ldf(afA, bfA, st, 2);
            __builtin_amdgcn_sched_barrier(0);
            mm(afB, bfB);
            __builtin_amdgcn_sched_barrier(0);
            ldf(afB, bfB, st, 3);
            __builtin_amdgcn_sched_barrier(0);
            mm(afA, bfA);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            ldf(afA, bfA, ((c + 1) & 1) * 3072, 0);
            __builtin_amdgcn_sched_barrier(0);
            mm(afB, bfB);
            __builtin_amdgcn_sched_barrier(0);
            loads(c);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 1.2345e301) out[(size_t)blockIdx.x * 512 + t] = s;
}

template <int V>
__global__ __launch_bounds__(512) void k_probe(double* out, const double* src, int chunks) {
    extern __shared__ double lds[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int i = t; i < 6144; i += 512) lds[i] = src[i] ;
    __syncthreads();
    d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
    const int wm = w & 3, wn = w >> 2;
    // conflict-free by construction (lane-contiguous; the tile gets there with its XOR swizzle): fragment f of k-step ks at [(2 ks + f) 64 + lane]
    const double* sA = lds + wm * 512 + lane;                                      // A: 4 x 512 doubles, B: 2 x 512, 3072 per stage
    const double* sB = lds + 2048 + wn * 512 + lane;
    double a0 = 1.0 + 1e-9 * lane, a1 = 1.0 - 1e-9 * lane, b0 = 0.5 + 1e-9 * lane, b1 = 0.5 - 1e-9 * lane;
    const double* gsrc = src + (size_t)(blockIdx.x % 64) * 4096 + w * 512 + lane * 2;
    for (int c = 0; c < chunks; ++c) {
        const int st = (c & 1) * 3072;
        if constexpr (V == 0) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
            }
        } else if constexpr (V == 4) {
            double af[4][2], bf[4][2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                af[ks][0] = sA[st + 128 * ks]; af[ks][1] = sA[st + 128 * ks + 64];
                bf[ks][0] = sB[st + 128 * ks]; bf[ks][1] = sB[st + 128 * ks + 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks][0], bf[ks][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks][0], bf[ks][1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks][1], bf[ks][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks][1], bf[ks][1], acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (V == 5) {
            double af[2], bf[2], an[2], bn[2];
            af[0] = sA[st]; af[1] = sA[st + 64]; bf[0] = sB[st]; bf[1] = sB[st + 64];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int kn = 128 * ((ks + 1) & 3);
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0], bf[0], acc[0][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                an[0] = sA[st + kn]; an[1] = sA[st + 64 + kn];
                __builtin_amdgcn_sched_barrier(0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0], bf[1], acc[0][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                bn[0] = sB[st + kn]; bn[1] = sB[st + 64 + kn];
                __builtin_amdgcn_sched_barrier(0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[1], bf[0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[1], bf[1], acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                af[0] = an[0]; af[1] = an[1]; bf[0] = bn[0]; bf[1] = bn[1];
            }
        } else if constexpr (V == 6) {
            // k-pairs adjacent in LDS: one 16-byte read per operand row covers two k-steps
            typedef double d2 __attribute__((ext_vector_type(2)));
            const d2* pA = reinterpret_cast<const d2*>(lds + st + wm * 512) + lane;
            const d2* pB = reinterpret_cast<const d2*>(lds + st + 2048 + wn * 512) + lane;
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const d2 x0 = pA[128 * kp], x1 = pA[128 * kp + 64], y0 = pB[128 * kp], y1 = pB[128 * kp + 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[h], y0[h], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[h], y1[h], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[h], y0[h], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[h], y1[h], acc[1][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // the tile's own order: the reads of step ks + 1 in front of the MFMAs of step ks, two register sets
            double af[2][2], bf[2][2];
            af[0][0] = sA[st]; af[0][1] = sA[st + 64]; bf[0][0] = sB[st]; bf[0][1] = sB[st + 64];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1, kn = 128 * ((ks + 1) & 3);
                __builtin_amdgcn_sched_barrier(0);
                af[nxt][0] = sA[st + kn]; af[nxt][1] = sA[st + 64 + kn]; bf[nxt][0] = sB[st + kn]; bf[nxt][1] = sB[st + 64 + kn];
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][0], bf[cur][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][0], bf[cur][1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][1], bf[cur][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][1], bf[cur][1], acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (V >= 3) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if constexpr (V >= 2) __syncthreads();
            if constexpr (V >= 3) {
                double* dst = lds + (c & 1) * 3072 + w * 384;                      // the stage just read, like the tile (its next reader is two barriers away)
#pragma unroll
                for (int h = 0; h < 3; ++h) __builtin_amdgcn_global_load_lds(gsrc + h * 128, dst + h * 128, 16, 0, 0);
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 1.2345e301) out[(size_t)blockIdx.x * 512 + t] = s;
}

template <int V>
static void run(const char* what, double* out, const double* src, int wg_per_cu, int lds_bytes) {
    const int chunks = 4096, grid = 256 * wg_per_cu * 4;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_probe<V>, dim3(grid), dim3(512), lds_bytes, 0, out, src, chunks);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double flops = (double)grid * 8 * chunks * 16 * 2048.0;
    printf("%-94s %d workgroups per CU: %6.2f TF/s  (%.3f of 78.6)\n", what, wg_per_cu, flops / best * 1e-9, flops / best * 1e-9 / 78.6);
    fflush(stdout);
}

// A loader wave (round 5 probe): 9 waves per workgroup.  Waves 0..7 run the loop of variant 2 (fragment reads, MFMAs, one barrier per chunk); wave 8 issues
// ALL 24 LDS-direct loads of the chunk after next (the eight waves' three slices each), waits for them and meets the others at the barrier.
template <int WRAP>
__global__ __launch_bounds__(576) void k_probe_loader(double* out, const double* A, int64_t lda, int tiles_m, int chunks) {
    extern __shared__ double lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int i = t; i < 6144; i += 576) lds[i] = 1.0 + 1e-6 * (i % 977);
    __syncthreads();
    const int bm = blockIdx.x % tiles_m, bn = (blockIdx.x / tiles_m) % (2 * tiles_m);
    const int lrow = lane >> 3, lg = lane & 7;
    if (w == 8) {
        const double* gA = A + (int64_t)(128 * bm + lrow) * lda + 2 * lg;          // + 8 q rows for slice q = 0..15
        const double* gB = A + (int64_t)(64 * bn + lrow) * lda + 2 * lg;           // + 8 q rows for slice q = 0..7
        for (int c = 0; c < chunks; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            double* dst = lds + (c & 1) * 3072;
            const int kc = 16 * ((c + 2) % WRAP);
#pragma unroll
            for (int q = 0; q < 16; ++q) __builtin_amdgcn_global_load_lds(gA + (int64_t)(8 * q) * lda + kc, dst + 128 * q, 16, 0, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) __builtin_amdgcn_global_load_lds(gB + (int64_t)(8 * q) * lda + kc, dst + 2048 + 128 * q, 16, 0, 0);
        }
        return;
    }
    d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
    const int wm = w & 3, wn = w >> 2;
    const double* sA = lds + wm * 512 + lane;
    const double* sB = lds + 2048 + wn * 512 + lane;
    for (int c = 0; c < chunks; ++c) {
        const int st = (c & 1) * 3072;
        double af[2][2], bf[2][2];
        af[0][0] = sA[st]; af[0][1] = sA[st + 64]; bf[0][0] = sB[st]; bf[0][1] = sB[st + 64];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1, kn = 128 * ((ks + 1) & 3);
            __builtin_amdgcn_sched_barrier(0);
            af[nxt][0] = sA[st + kn]; af[nxt][1] = sA[st + 64 + kn]; bf[nxt][0] = sB[st + kn]; bf[nxt][1] = sB[st + 64 + kn];
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][0], bf[cur][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][0], bf[cur][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][1], bf[cur][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][1], bf[cur][1], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 1.2345e301) out[(size_t)blockIdx.x * 512 + t] = s;
}

template <int WRAP>
static void run_loader(const char* what, double* out, const double* A, int64_t lda) {
    const int chunks = 2048, grid = 256 * 3 * 4, lds_bytes = 52 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe_loader<WRAP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_probe_loader<WRAP>, dim3(grid), dim3(576), lds_bytes, 0, out, A, lda, 62, chunks);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double flops = (double)grid * 8 * chunks * 16 * 2048.0;
    printf("%-94s 3 workgroups per CU: %6.2f TF/s  (%.3f of 78.6)\n", what, flops / best * 1e-9, flops / best * 1e-9 / 78.6);
    fflush(stdout);
}

template <int WRAP, int MODE = 0>
static void run_mem(const char* what, double* out, const double* A, int64_t lda) {
    const int chunks = 2048, grid = 256 * 3 * 4, lds_bytes = 52 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe_mem<WRAP, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_probe_mem<WRAP, MODE>), dim3(grid), dim3(512), lds_bytes, 0, out, A, lda, 62, chunks, 0);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double flops = (double)grid * 8 * chunks * 16 * 2048.0;
    printf("%-94s 3 workgroups per CU: %6.2f TF/s  (%.3f of 78.6)\n", what, flops / best * 1e-9, flops / best * 1e-9 / 78.6);
    fflush(stdout);
}

int main() {
    double *out, *src;
    CHECK(hipMalloc(&out, (size_t)256 * 3 * 4 * 512 * 8));
    CHECK(hipMalloc(&src, (size_t)64 * 4096 * 8 + 65536));
    std::vector<double> h(64 * 4096 + 8192);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0 + 1e-6 * (double)(i % 977);
    CHECK(hipMemcpy(src, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    const int L3 = 52 * 1024, L1 = 150 * 1024, L2 = 76 * 1024;
    run<0>("0 MFMAs only", out, src, 3, L3);
    run<1>("1 + fragment reads (LDS), the tile's order", out, src, 3, L3);
    run<1>("1 + fragment reads (LDS), the tile's order", out, src, 2, L2);
    run<1>("1 + fragment reads (LDS), the tile's order", out, src, 1, L1);
    run<2>("2 + a barrier per chunk of 16 MFMAs", out, src, 3, L3);
    run<3>("3 + three LDS-direct loads per wave and chunk, vmcnt(0) before the barrier", out, src, 3, L3);
    run<4>("4 = 1 with a chunk's 8 fragment reads first, then its 16 MFMAs", out, src, 3, L3);
    run<5>("5 = 1 with the next step's reads issued between the MFMAs", out, src, 3, L3);
    run<6>("6 = 1 with 16-byte fragment reads (k-pairs adjacent): half the LDS instructions", out, src, 3, L3);
    run<0>("0 MFMAs only", out, src, 1, L1);
    // the tile's real operand addressing on an 8192 x 8208 matrix (row stride 64 KiB + 128 B, as the workspaces)
    double* A;
    const int64_t lda = 8208;
    CHECK(hipMalloc(&A, (size_t)8192 * lda * 8));
    CHECK(hipMemset(A, 0, (size_t)8192 * lda * 8));
    run_mem<1>("7 = 3 with the tile's operand addressing, every chunk the same 24 KB per workgroup (L2-hot)", out, A, lda);
    run_mem<8>("7 ... a K = 128 walk per workgroup, wrapped (mostly L2)", out, A, lda);
    run_mem<128>("7 ... a K = 2048 walk per workgroup, wrapped (Infinity Cache / HBM)", out, A, lda);
    run_mem<512>("7 ... a K = 8192 walk per workgroup, wrapped", out, A, lda);
    run_mem<128, 1>("8 = 7 (K = 2048 walk) + a wave-uniform branch around every group of four MFMAs (the tile's idle-wave test)", out, A, lda);
    run_mem<128, 2>("9 = 7 in the tile's order: 12 MFMAs, wait + barrier, next chunk's first reads, 4 MFMAs, the loads", out, A, lda);
    run_mem<128, 3>("10 = 8 + 9: the shipped loop without its C phases", out, A, lda);
    // random operands instead of zeros (board power: the clock under real data)
    {
        std::vector<double> hA((size_t)8192 * lda);
        unsigned long long z = 88172645463325252ull;
        for (size_t i = 0; i < hA.size(); ++i) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; hA[i] = (double)(long long)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0; }
        CHECK(hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
    }
    run_mem<128, 3>("10 on random operands", out, A, lda);
    run_mem<128, 0>("7 on random operands", out, A, lda);
    run_loader<128>("12 = a NINTH wave issues all 24 LDS-direct loads of a chunk, the eight others only read fragments and multiply; random operands", out, A, lda);
    run_mem<128, 4>("11 = 7 with the operands through registers (global_load_dwordx4 a chunk ahead, ds_write_b128, two barriers per chunk), random operands", out, A, lda);
    return 0;
}
