#!/usr/bin/env python3
"""Single-factorisation latency at n = 8192: the default look-ahead schedule and each option changed on its own."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.lab_context(0)
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
DEFAULTS = {"bulk_lds_pad": 80 * 1024, "la_depth2": 1, "chain_prefetch": 1, "reserve_cus": 0, "la_split": 0, "chain_window": 0,
            "chain_fused": -1}
VARIANTS = [("defaults", {}), ("bulk_lds_pad=0 (three bulk workgroups per CU)", {"bulk_lds_pad": 0}), ("la_depth2=0", {"la_depth2": 0}),
            ("bulk_lds_pad=0 la_depth2=0", {"bulk_lds_pad": 0, "la_depth2": 0}), ("chain_prefetch=0", {"chain_prefetch": 0}),
            ("reserve_cus=2", {"reserve_cus": 2}), ("la_split=2048", {"la_split": 2048}), ("chain_window=1", {"chain_window": 1}),
            ("chain_fused=1", {"chain_fused": 1}), ("chain_fused=1 chain_window=1", {"chain_fused": 1, "chain_window": 1})]
for rep in range(3):
    for label, opts in VARIANTS:
        for k, v in {**DEFAULTS, **opts}.items():
            ctx.set_option(k, v)
        run(label)
for k, v in DEFAULTS.items():
    ctx.set_option(k, v)
ctx.set_option("lookahead", 0)
run("lookahead=0")
