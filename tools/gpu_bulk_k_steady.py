#!/usr/bin/env python3
"""Steady-state rate of the bulk tile (k_gemm_ld3, cfg 7) as a function of K: `reps` back-to-back lower-triangular launches of the
first outer step's shape on device-resident operands, chip full throughout.  K = 256 moves one C tile per 256 columns of products,
K = 512 half as many C bytes per flop, K = 1024 a quarter: if the rate does not rise with K, C traffic is not what holds the kernel
at 55 of the 66 TF/s its K loop reaches, and no way of hiding or halving the C phase will move the batch rate."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402

ctx = gsum_amd.lab_context(0)
M = 7936
for rnd in range(2):
    for K in (256, 512, 1024, 2048):
        for reps in (3, 200):
            tf, us = ctx.bench_gemm_nt(7, M, M, K, tri=True, lda=8192 + 16, reps=reps)
            print(f"round {rnd} M={M} K={K:5d} reps={reps:3d}: {tf:6.2f} TFLOP/s  {us:8.1f} us per launch", flush=True)
