#!/usr/bin/env python3
"""Isolated launches of the MFMA tile kernel for rocprofv3 --pmc:  prof_gemm.py cfg M K tri reps"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402

cfg, M, K, tri, reps = (int(a) for a in (sys.argv[1:] + ["5", "8192", "256", "1", "3"])[:5])
ctx = gsum_amd.lab_context(0)
print(cfg, M, K, tri, ctx.bench_gemm_nt(cfg, M, M, K, bool(tri), 8208, reps), flush=True)
