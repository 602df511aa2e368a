#!/usr/bin/env python3
"""A/B of the bulk tile's LDS stage count (option bulk_stages = 2 | 3) in ONE process, interleaved rounds: exclusive SYRK rates,
pipelined evaluation throughput (16 in flight) and one factorisation alone, n = 8192; results must be bit-identical."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.lab_context(0)
ctx.bench_gemm_nt(7, 7936, 7936, 256, True, 8208)
n = 8192
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
desc = gsum_amd.describe_kernel(RBF(0.2), 1)
ctx.set_inputs(X, Z)
ctx.lml_resident([desc] * 16, 1e-10)
ref = None
for rnd in range(3):
    for st in (2, 3):
        ctx.set_option("bulk_stages", st)
        res = {}
        for name, args in (("syrk8k", (7, 7936, 7936, 256, True, 8208)), ("syrk8k_k512", (7, 7936, 7936, 512, True, 8208)),
                           ("syrk4k", (7, 4096, 4096, 256, True, 8208)), ("syrk2k", (7, 2048, 2048, 256, True, 8208))):
            res[name] = round(float(np.median([ctx.bench_gemm_nt(*args)[0] for _ in range(3)])), 2)
        ctx.set_option("batch_slots", 16)
        ctx.lml_resident([desc] * 16, 1e-10)
        t0 = time.perf_counter()
        G, sld, info = ctx.lml_resident([desc] * 32, 1e-10)
        res["evals_per_s_16_in_flight"] = round(32 / (time.perf_counter() - t0), 1)
        ctx.set_option("batch_slots", 1)
        ts = []
        for _ in range(4):
            ctx.lml_resident([desc], 1e-10)
            ts.append(ctx.timers()["potrf_ms"])
        res["single_potrf_ms"] = round(min(ts), 3)
        key = (float(sld[0]).hex(), float(G[0, 0, 0]).hex())
        ref = ref or key
        res["identical"] = key == ref
        print("stages", st, res, flush=True)
