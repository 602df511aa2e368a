/*
 * gsum_hip_debug.h -- the LAB side of libgsum_hip: diagnostics, probes, microbenchmarks and the schedule / kernel-generation
 * switches the measurements of DESIGN.md were made with.  None of it is part of the drop-in contract (include/gsum_hip.h): these
 * entry points and option names exist only in libgsum_hip_lab.so, the same sources compiled with -DGSUM_LAB
 * (python -m gsum_amd.build --lab; tools/ and the schedule-equivalence tests load it).  Every switch leaves results bit-identical.
 */
#ifndef GSUM_HIP_DEBUG_H
#define GSUM_HIP_DEBUG_H

#include "gsum_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- options accepted by gsum_set_option only in the lab build ---------------------------------------------------------------
 * batch schedule: "wave_min" (calls with fewer evaluations run them one after the other), "wave_depth" (panels per far update,
 *   default 4), "wave_deep_rows", "wave_near_on_chain", "wave_serial", "wave_shift", "wave_panel_wg4" (0 | 4 | 8 waves per panel workgroup), "lazy_far" (0: no grouping of trailing
 *   updates), "lazy_min_np";  gradient batches: "batch_slots", "grad_batch_wave";  one gradient evaluation alone: "grad_interleave" (the sweep's launches
 *   enqueued between the factorisation's outer steps), "grad_split" (Q_p beside the R^-1 product, traces from stored dR triangles), "grad_lazy_chain"
 * single factorisation: "chain_rows" 256 | 512, "chain_lazy", "chain_min_np", "chain_fused", "chain_prefetch", "la_depth2",
 *   "bulk_lds_pad", "chain_test_abort" (one-shot give-up of the persistent chain at that outer step), "chain_stamps"
 * other sweeps: "predict_lazy", "medium_lazy", "build_lower_only", "diag_stamps", "panel_stats", "bench_fill" */

/* diagnostic: the 64 raw stamp words of the last diagonal-block kernel run with "diag_stamps" = 1 ([0..4] as above;
 * round-2 kernel: [7] start, [8 + 2j], [9 + 2j] wave 0 behind the two barriers of micro-block column j, [24 + j] cycles
 * of the pivot recurrence of micro-block j). */
int gsum_debug_diag_stamps(gsum_ctx* ctx, int64_t* out64);
/* the pairwise stream probe of gsum_init (option "pipes_ok") on the context's four streams plus `extra` (0..4) streams created for the
 * call: out = (4 + extra)^2 overlaps of the pairs' two 100-us kernels in 1/1000 of their length (diagonal 1000).  A pair on one
 * command-processor pipe takes turns (~0); DESIGN.md section 4.1. */
int gsum_debug_pipe_probe(gsum_ctx* ctx, int32_t extra, int32_t* out);
/* diagnostic: per-outer-step realtime stamps of the last persistent-chain factorisation (options "chain_stamps" = 1,
 * "chain_persist"; gsum_potrf_lower / a single fused evaluation on a matrix whose order is a multiple of 256): out holds
 * 24 values per step in 100 MHz ticks relative to the first stamp (-1: not written); [16..23] = first start / last end of the
 * step's four host-enqueued launches (panel of the rows below the window, updates A, B, Far).  Indices: D role 0 step begins, 1 its
 * diagonal block is up to date, 2 first block's tables published, 3 block row k + 1 up to date, 4 L(k+1, k) published,
 * 5 sibling update done, 6 second block's tables published; P wave 0: 8 its rows are up to date, 9 first tables seen,
 * 10 sibling update done, 11 second tables seen, 12 rows published, 13 its first update task starts, 14 is done.
 * This is the timeline evidence for numpy.linalg.cholesky at gsum/models.py:711, 809, 969 (one factorisation alone). */
int gsum_debug_chain_stamps(gsum_ctx* ctx, double* out, int32_t max_steps, int32_t* steps);
/* fp64 MFMA issue-rate probe (v_mfma_f64_16x16x4_f64, operands in registers, waves_per_simd resident
 * waves on every SIMD, n_acc independent accumulators per wave; n_acc = 1 gives the dependent latency):
 * out3 = {achieved TFLOP/s, shader cycles per MFMA per wave, in-kernel clock GHz}. */
int gsum_probe_mfma_f64(gsum_ctx* ctx, int32_t iters, int32_t waves_per_simd, int32_t n_acc, double* out3);
/* HBM streaming-store probe: achieved GB/s writing `bytes` with 16-B stores. */
int gsum_probe_hbm_write(gsum_ctx* ctx, int64_t bytes, double* gbps);
/* microbenchmark of the MFMA tile kernel on device-resident pseudo-random operands (leading dimension
 * lda >= K for A and B, as inside the factorisation): out2 = {algorithmic TFLOP/s, microseconds per launch}.
 * tri != 0: SYRK form (B = A, lower tiles only, M == N, flops counted as M(M+1)K).
 * Option "bench_fill" = 1 zeroes the operands first (timing is value-independent, board power is not: tools/gpu_power_probe.py).
 * cfg = 99 is not a GEMM: the pure issue rate of v_mfma_f64_16x16x4_f64 on register operands -- M workgroups of N threads (a multiple
 * of 64, <= 512), K rounds of lda (4 or 8) independent MFMAs per wave, tri ignored: what the matrix pipes sustain on this card
 * (77.6 TFLOP/s measured, profiles/r03_mfma_peak.log), the ceiling the tile kernels are measured against. */
int gsum_bench_gemm_nt(gsum_ctx* ctx, int32_t cfg, int32_t tri, int64_t M, int64_t N, int64_t K, int64_t lda,
                       int32_t reps, double* out2);
/* debug: C(MxN) = beta*C + sign * A(MxK) B(NxK)^T through the MFMA tile kernel (cfg
 * 1: 32x128 tile, 2: 16x256 tile, 5: 128x128 tile with 8 waves (register staging), 7: 128x64
 * tile with LDS-direct staging, three workgroups per CU; tri != 0: lower tiles only, needs M == N). */
int gsum_debug_gemm_nt(gsum_ctx* ctx, int32_t cfg, int32_t tri, double* C, const double* A, const double* B,
                       int64_t M, int64_t N, int64_t K, int32_t beta, double sign);


#ifdef __cplusplus
}
#endif
#endif /* GSUM_HIP_DEBUG_H */
