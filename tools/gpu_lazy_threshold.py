#!/usr/bin/env python3
"""From which order on the batch schedule's K = 512 far updates (lazy_far = 2: no more launches than the plain schedule) pay: pipelined
throughput at several n with the threshold option lazy_min_np above every order ("plain") and below every order ("lazy"), same process,
interleaved.  Usage: gpu_lazy_threshold.py [n ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.lab_context(0)
desc = gsum_amd.describe_kernel(RBF(0.2), 1)
for n in [int(a) for a in sys.argv[1:]] or (4352, 5120, 6144, 7168, 8192, 12288):
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
    ctx.set_inputs(X, Z)
    ctx.set_option("batch_slots", 20)
    ref = None
    out = []
    for rnd in range(2):
        for thr in (32768, 1024):
            ctx.set_option("lazy_min_np", thr)
            ctx.lml_resident([desc] * 20, 1e-10)
            rates = []
            for _ in range(2):
                t0 = time.perf_counter()
                G, sld, info = ctx.lml_resident([desc] * 40, 1e-10)
                rates.append(40 / (time.perf_counter() - t0))
            key = (float(sld[0]).hex(), float(G[-1, 2, 3]).hex())
            ref = ref or key
            out.append(f"{'lazy' if thr == 1024 else 'plain'} {max(rates):7.1f}{'' if key == ref else ' DIFFERENT'}")
    print(f"n={n:6d}: " + "   ".join(out), flush=True)
ctx.set_option("lazy_min_np", 4352)
