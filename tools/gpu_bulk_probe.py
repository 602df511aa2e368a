#!/usr/bin/env python3
"""Where the bulk tile's K loop loses its 15 %: 200 back-to-back lower-triangular launches (M = 7936) with the C read (1), the C store (2)
removed and / or every tile fetching the operand rows of tile 0 (4: a 256-KB working set that lives in the L2, i.e. the K loop without
operand-fetch latency from the Infinity Cache / HBM).  Probe code: option bulk_probe (results are wrong by design while it is set)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import gsum_amd  # noqa: E402

ctx = gsum_amd.default_context(0)
M = 7936
ctx.bench_gemm_nt(7, M, M, 256, tri=True, lda=8208, reps=50)
names = {0: "as shipped", 1: "no C read", 2: "no C store", 3: "no C read, no C store", 4: "operands L2-hot", 7: "no C phases, operands L2-hot",
         5: "no C read, operands L2-hot"}
for rnd in range(2):
    for probe in (0, 3, 7, 4, 1, 2):
        ctx.set_option("bulk_probe", probe)
        row = []
        for K in (256, 512, 2048):
            tf, us = ctx.bench_gemm_nt(7, M, M, K, tri=True, lda=8208, reps=200 if K < 2048 else 50)
            row.append(f"K={K}: {tf:5.1f} TF/s {us:7.1f} us")
        print(f"round {rnd} probe {probe} ({names[probe]:30s})  " + "   ".join(row), flush=True)
ctx.set_option("bulk_probe", 0)
