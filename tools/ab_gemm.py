#!/usr/bin/env python3
"""One leg of an A/B comparison (library chosen by $GSUM_HIP_LIBRARY): exclusive rates of the bulk kernel and
pipelined evaluation throughput at n = 8192.  Run alternately for the two libraries."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.lab_context(0)
tag = os.environ.get("AB_TAG", "?")
ctx.bench_gemm_nt(7, 7936, 7936, 256, True, 8208)           # warm-up (first measurement in a process reads low)
res = {}
for name, args in (("syrk8k", (7, 7936, 7936, 256, True, 8208)), ("syrk4k", (7, 4096, 4096, 256, True, 8208)),
                   ("gemm4k", (7, 4096, 4096, 256, False, 8208)), ("trsm", (1, 7936, 128, 128, False, 8208)),
                   ("la", (1, 7936, 256, 256, False, 8208))):
    vals = [ctx.bench_gemm_nt(*args)[0] for _ in range(3)]
    res[name] = round(float(np.median(vals)), 2)
n = 8192
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
desc = gsum_amd.describe_kernel(RBF(0.2), 1)
ctx.set_inputs(X, Z)
ctx.lml_resident([desc] * 20, 1e-10)
t0 = time.perf_counter()
G, sld, info = ctx.lml_resident([desc] * 40, 1e-10)
res["ms_per_eval_pipelined"] = round((time.perf_counter() - t0) / 40 * 1e3, 3)
ctx.set_option("batch_slots", 1)
ctx.lml_resident([desc], 1e-10)
t0 = time.perf_counter()
ctx.lml_resident([desc] * 4, 1e-10)
res["ms_single"] = round((time.perf_counter() - t0) / 4 * 1e3, 3)
res["sld"] = float(sld[0]).hex()
res["g00"] = float(G[0, 0, 0]).hex()
print(tag, res, flush=True)
