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
    return 0;
}
