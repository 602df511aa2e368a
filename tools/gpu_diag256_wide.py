#!/usr/bin/env python3
"""A/B of the fused diagonal kernel's register budget in the batch regime (option diag256_wide: k_potrf_diag256 at 256 registers with
228 B/lane of spills against the same code with the whole register file, 256 + 176, no spills), one process, interleaved rounds:
pipelined evaluation throughput at n = 8192 with 20 in flight; results must be bit-identical."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.default_context(0)
n = 8192
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
desc = gsum_amd.describe_kernel(RBF(0.2), 1)
ctx.set_inputs(X, Z)
ctx.set_option("batch_slots", 20)
ctx.lml_resident([desc] * 20, 1e-10)
ref = None
for rnd in range(4):
    for wide in (0, 1):
        ctx.set_option("diag256_wide", wide)
        ctx.lml_resident([desc] * 20, 1e-10)
        rates = []
        for _ in range(3):
            t0 = time.perf_counter()
            G, sld, info = ctx.lml_resident([desc] * 40, 1e-10)
            rates.append(40 / (time.perf_counter() - t0))
        key = (float(sld[0]).hex(), float(G[0, 0, 0]).hex(), float(G[-1, 2, 3]).hex())
        ref = ref or key
        print(f"round {rnd} diag256_wide={wide}: {np.median(rates):6.1f} evals/s (20 in flight; {min(rates):.1f} .. {max(rates):.1f}), "
              f"identical {key == ref}", flush=True)
