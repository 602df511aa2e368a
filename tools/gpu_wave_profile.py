#!/usr/bin/env python3
"""Per-class HIP-event times of one batch call (20 evaluations, n = 8192) for several group layouts: with ONE group nothing runs
beside the trailing updates, so the difference in the bulk launches' rate is what the chain kernels' co-residency costs them.
    python tools/gpu_wave_profile.py [K]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, r = 8192, 6
X = 0.1 * np.arange(n)[:, None]
c = np.random.RandomState(0).randn(n, r)
Z = np.concatenate([c, np.ones((n, 1))], axis=1)
ctx = gsum_amd.lab_context(0)
ctx.set_inputs(X, Z)
descs = ctx.desc_array([gsum_amd.describe_kernel(RBF(float(e)), 1) for e in np.linspace(0.19, 0.21, K)])
for groups, size, depth, serial, wg4 in ((3, 8, 4, 0, 4), (1, 20, 4, 0, 4), (3, 8, 4, 0, 8), (3, 8, 4, 0, 0), (3, 8, 4, 1, 4), (3, 8, 4, 0, 4), (1, 20, 4, 0, 4)):
    ctx.set_option("wave_panel_wg4", wg4)
    ctx.set_option("wave_serial", serial)
    ctx.set_option("wave_groups", groups)
    ctx.set_option("wave_size", size)
    ctx.set_option("wave_depth", depth)
    ctx.lml_resident(descs, 1e-10)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.lml_resident(descs, 1e-10)
        ts.append(time.perf_counter() - t0)
    ctx.set_option("profile_gemm", 1)
    ctx.kernel_profile()
    t0 = time.perf_counter()
    ctx.lml_resident(descs, 1e-10)
    wall = time.perf_counter() - t0
    prof = ctx.kernel_profile()
    ctx.set_option("profile_gemm", 0)
    b = prof["bulk_update"]
    print(json.dumps({"groups": groups, "size": size, "depth": depth, "serial": serial, "wg4": wg4, "evals_per_s": K / float(np.median(ts)), "profiled_ms": wall * 1e3,
                      "bulk_ms": b["ms"], "bulk_launches": b["launches"], "bulk_tflops": b["flops"] / b["ms"] / 1e9,
                      "panel_ms": prof["panel_gemm"]["ms"], "diag_ms": prof["diag_block"]["ms"], "build_ms": prof["kernel_build"]["ms"]}), flush=True)
