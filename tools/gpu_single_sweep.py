#!/usr/bin/env python3
"""Single-factorisation latency at n = 8192 against the knobs that shape the look-ahead schedule."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.default_context(0)
n, r = 8192, 6
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, r), np.ones((n, 1))], axis=1)
desc = gsum_amd.describe_kernel(RBF(0.2), 1)
ctx.set_inputs(X, Z)
ctx.set_option("batch_slots", 1)


def run(label):
    ts = []
    for _ in range(6):
        ctx.lml_resident([desc], 1e-10)
        ts.append(ctx.timers()["potrf_ms"])
    print(f"{label:40s} potrf ms min {min(ts):.3f} median {sorted(ts)[3]:.3f}", flush=True)


ctx.set_option("chain_fused", 0)
ctx.set_option("chain_window", 0)
for rep in range(3):
    for split in (0, 1024, 2048, 3072, 4096):
        ctx.set_option("la_split", split)
        run(f"la_split={split}")
ctx.set_option("la_split", 0)
ctx.set_option("bulk_lds_pad", 0)
ctx.set_option("reserve_cus", 0)
ctx.set_option("bulk_cfg", 7)
ctx.set_option("lookahead", 0)
run("lookahead=0")
