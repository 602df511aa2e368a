#!/usr/bin/env python3
"""Throughput of the grouped batch schedule for a list of settings (lab library).
       python tools/gpu_wave_sweep.py "K=20" "K=20,wave_groups=2,wave_size=10" "K=84,wave_shift=4" ...
   every item starts from the defaults (3 x 7, depth 4, four-wave panel workgroups, groups in phase)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402
from gpu_wave_check import workload  # noqa: E402

DEFAULTS = {"wave_groups": 3, "wave_size": 8, "wave_depth": 4, "wave_deep_rows": 3072, "wave_shift": 0, "wave_panel_wg4": 4,
            "wave_near_on_chain": 1, "wave_serial": 0, "wave_head": 124, "wave_panel_rows_lds": 1}
n = int(os.environ.get("N", "8192"))
ctx = gsum_amd.lab_context(0)
X, Z = workload(n, 6)
ctx.set_inputs(X, Z)
for item in sys.argv[1:]:
    opts = dict(DEFAULTS)
    K, reps = 20, 7
    for kv in item.split(","):
        k, v = kv.split("=")
        if k == "K":
            K = int(v)
        elif k == "reps":
            reps = int(v)
        else:
            opts[k] = int(v)
    for k, v in opts.items():
        ctx.set_option(k, v)
    descs = ctx.desc_array([gsum_amd.describe_kernel(RBF(float(e)), 1) for e in np.linspace(0.19, 0.21, K)])
    ctx.lml_resident(descs, 1e-10)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.lml_resident(descs, 1e-10)
        ts.append(time.perf_counter() - t0)
    print(json.dumps({"item": item, "evals_per_s_median": round(K / float(np.median(ts)), 1), "best": round(K / min(ts), 1),
                      "ms_per_call": round(float(np.median(ts)) * 1e3, 2)}), flush=True)
