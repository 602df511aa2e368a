#!/usr/bin/env python3
"""Pipelined evaluation throughput at n = 8192 against the number of evaluations in flight (option batch_slots), in steady state (96
evaluations per timing, so ramp and tail are a small share): does the chip still have idle time that more independent streams would fill?"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[1] if len(sys.argv) > 1 else "40")
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.lab_context(0)
n = 8192
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
desc = gsum_amd.describe_kernel(RBF(0.2), 1)
ctx.set_inputs(X, Z)
print("GPU_MAX_HW_QUEUES", os.environ["GPU_MAX_HW_QUEUES"], flush=True)
ref = None
for rnd in range(2):
    for slots in (12, 16, 20, 24, 28, 32):
        ctx.set_option("batch_slots", slots)
        got = ctx.get_option("batch_slots")
        ctx.lml_resident([desc] * slots, 1e-10)
        rates = []
        for _ in range(2):
            t0 = time.perf_counter()
            G, sld, info = ctx.lml_resident([desc] * 96, 1e-10)
            rates.append(96 / (time.perf_counter() - t0))
        key = (float(sld[0]).hex(), float(G[-1, 2, 3]).hex())
        ref = ref or key
        print(f"round {rnd} slots asked {slots:2d} got {got:2d}: {max(rates):6.1f} evals/s (other run {min(rates):6.1f}), identical {key == ref}", flush=True)
