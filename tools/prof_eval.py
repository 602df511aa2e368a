#!/usr/bin/env python3
"""A few single evaluations (one factorisation alone, look-ahead schedule) at the given orders, for rocprofv3.
Prints which schedule ran: under a tool that serialises dispatches of different streams the persistent-chain schedule is
switched off by the library's two-stream probe (chain_probe = -1) and the host-enqueued schedule is what the trace shows."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.lab_context(0)
ctx.set_option("batch_slots", 1)
if os.environ.get("GSUM_PROF_CHAIN") is not None:
    ctx.set_option("chain_persist", int(os.environ["GSUM_PROF_CHAIN"]))
for n in [int(a) for a in sys.argv[1:]] or [8192]:
    r = 6 if n >= 8192 else 4
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, r), np.ones((n, 1))], axis=1)
    ctx.set_inputs(X, Z)
    desc = gsum_amd.describe_kernel(RBF(0.2), 1)
    for _ in range(5):
        ctx.lml_resident([desc], 1e-10)
    print(n, ctx.timers(), "chain_probe", ctx.get_option("chain_probe"), "chain_aborts", ctx.get_option("chain_aborts"),
          "chain_persist", ctx.get_option("chain_persist"), flush=True)
