// probes
// (part of gsum_kernels.hip.h: included from there, in order; gfx950 only)
#pragma once
// ---- probes ------------------------------------------------------------------------------------
// pseudo-random fill in [-1, 1) (integer hash), so benchmark operands are not zeros (DVFS reads high on zeros)
__global__ __launch_bounds__(256) void k_fill_random(double* p, int64_t n, unsigned seed) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = (double)(long long)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;
    }
}

// NACC independent accumulators held in VGPRs (inline asm: hipcc would otherwise shuttle them through
// AGPRs every iteration), back-to-back v_mfma_f64_16x16x4_f64.  NACC = 1 measures dependent latency.
template <int NACC>
__global__ __launch_bounds__(256) void k_probe_mfma(double* out, int iters, unsigned long long* stamps) {
    gs_d4 acc[NACC];
#pragma unroll
    for (int u = 0; u < NACC; ++u) acc[u] = (gs_d4){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16 / NACC; ++rep)
#pragma unroll
            for (int u = 0; u < NACC; ++u)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
    }
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < NACC; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[(int64_t)blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {     // diagnostic stamps go to their own buffer, never into results
        const int64_t wv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * wv] = c1 - c0;
        stamps[2 * wv + 1] = r1 - r0;
    }
}

__global__ __launch_bounds__(256) void k_probe_store(gs_d2* out, int64_t nvec) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    gs_d2 v = {1.0, 2.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) out[i] = v;
}

// one wave that stamps its start, spins for `ticks` of the 100 MHz real-time counter and stamps its end: the pairwise stream probe of
// gsum_init (gs_pipe_probe): two such kernels on two streams overlap in time iff the streams' queues are served side by side
__global__ __launch_bounds__(64) void k_probe_stamp(unsigned long long ticks, unsigned long long* out2) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) {
        out2[0] = t0;
        out2[1] = __builtin_amdgcn_s_memrealtime();
    }
}

// one wave spinning for `ticks` of the 100 MHz real-time counter: the queue-concurrency probe (gs_probe_queues)
__global__ __launch_bounds__(64) void k_probe_spin(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}


