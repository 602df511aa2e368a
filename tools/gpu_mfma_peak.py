#!/usr/bin/env python3
"""What the fp64 matrix pipes sustain on this card with nothing but MFMAs in flight (k_mfma_peak: register operands, 4 or 8 independent
accumulators per wave, no LDS, no memory, no barrier), against the number of waves per SIMD: the ceiling the bulk tile's K loop
(64.4 TF/s with its C phases removed) should be measured against."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402

ctx = gsum_amd.lab_context(0)
for rnd in range(2):
    for nacc in (4, 8):
        for threads, wgs_per_cu in ((256, 1), (512, 1), (512, 2), (512, 3), (512, 4)):
            waves = threads // 64 * wgs_per_cu / 4
            iters = 20000 // nacc * 4 // max(1, int(waves))
            tf, us = ctx.bench_gemm_nt(99, 256 * wgs_per_cu, threads, iters, lda=nacc, reps=5)
            print(f"round {rnd} {nacc} accumulators/wave, {waves:3.0f} waves/SIMD ({wgs_per_cu} x {threads} threads per CU): {tf:6.2f} TFLOP/s "
                  f"= {tf / 78.6:5.3f} of 78.6   ({us:8.1f} us per launch)", flush=True)
