#!/usr/bin/env python3
"""Hardware probes + stage timings on the GPU box; writes gpurun_out/probe.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

out = {}
ctx = gsum_amd.lab_context(0)
out["mfma_f64"] = {f"wps{w}_acc{a}": ctx.probe_mfma_f64(4000, w, a)
                   for w, a in ((1, 1), (1, 2), (1, 4), (1, 8), (1, 16), (2, 8), (2, 16), (4, 4), (4, 8), (8, 4), (1, 16))}
for k, v in out["mfma_f64"].items():
    print(k, {a: round(b, 3) for a, b in v.items()}, flush=True)
out["hbm_write_gbps"] = [ctx.probe_hbm_write(1 << 30) for _ in range(3)]
print(out, flush=True)
desc = gsum_amd.describe_kernel(RBF(0.2), 1)
ctx.set_option("diag_stamps", 1)
for n, r in ((128, 4), (2048, 4), (4096, 6), (8192, 6)):
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, r), np.ones((n, 1))], axis=1)
    ctx.set_inputs(X, Z)
    for la in (1, 0):
        for lower in (1, 0):
            ctx.set_option("lookahead", la)
            ctx.set_option("build_lower_only", lower)
            ts = []
            for rep in range(5):
                t0 = time.perf_counter()
                G, sld, info = ctx.lml_resident([desc], 1e-10)
                wall = (time.perf_counter() - t0) * 1e3
                tm = ctx.timers()
                tm["wall_ms"] = wall
                ts.append(tm)
            best = min(ts, key=lambda t: t["total_ms"])
            if n == 2048:
                print("diag stamps", ctx.diag_stamps(), flush=True)
            best["chol_tflops"] = n ** 3 / 3 / (best["potrf_ms"] * 1e-3) / 1e12
            best["build_gbps_8n2"] = 8.0 * n * n / (best["build_ms"] * 1e-3) / 1e9
            out[f"n{n}_la{la}_lower{lower}"] = best
            print(n, la, lower, {k: round(v, 4) for k, v in best.items()}, int(info[0]), flush=True)
ctx.set_option("lookahead", 1)
ctx.set_option("build_lower_only", 1)
# batched evaluations, several in flight
n, r = 8192, 6
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, r), np.ones((n, 1))], axis=1)
ctx.set_inputs(X, Z)
G_ref = None
for la in (1, 0):
    ctx.set_option("lookahead", la)
    for slots in (1, 3, 10, 14, 20):
        ctx.set_option("batch_slots", slots)
        ctx.lml_resident([desc] * slots, 1e-10)
        nb = 18
        t0 = time.perf_counter()
        G, sld, info = ctx.lml_resident([desc] * nb, 1e-10)
        dt = time.perf_counter() - t0
        G_ref = G[0] if G_ref is None else G_ref
        key = f"batch_la{la}_slots{slots}"
        out[key] = dict(ms_per_eval=dt / nb * 1e3, evals_per_s=nb / dt, chol_tflops_aggregate=n ** 3 / 3 * nb / dt / 1e12,
                        bit_identical=bool((G == G_ref).all()))
        print(key, out[key], flush=True)
ctx.set_option("lookahead", 1)
ctx.set_option("batch_slots", 20)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "probe.json"), "w") as f:
    json.dump(out, f, indent=1)
