#!/usr/bin/env python3
"""How much of the chip's register file the panel kernel holds, and for how long: sum of the lifetimes of all k_panel256 waves (option
panel_stats, in-kernel s_memrealtime) in the pipelined batch (20 in flight, n = 8192) and for one factorisation at a time.
footprint = sum of wave lifetimes x 224 registers / (1024 SIMDs x 512 registers x wall time)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.lab_context(0)
n = 8192
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
desc = gsum_amd.describe_kernel(RBF(0.2), 1)
ctx.set_inputs(X, Z)
for rnd in range(2):
    for name, slots, count in (("batch, 20 in flight", 20, 80), ("batch, 12 in flight", 12, 48), ("one factorisation at a time", 1, 20)):
        ctx.set_option("batch_slots", slots)
        ctx.lml_resident([desc] * slots, 1e-10)
        ctx.set_option("panel_stats", 1)
        t0 = time.perf_counter()
        ctx.lml_resident([desc] * count, 1e-10)
        wall = time.perf_counter() - t0
        ticks, waves = ctx.get_option("panel_wave_ticks"), ctx.get_option("panel_waves")
        ctx.set_option("panel_stats", 0)
        life_us = ticks / 100.0 / max(waves, 1)
        foot = ticks * 1e-8 * 224 / (1024 * 512 * wall)
        print(f"round {rnd} {name:28s}: {count / wall:6.1f} evals/s, {waves / count:7.1f} panel waves per evaluation, mean wave lifetime {life_us:7.1f} us, "
              f"panel register-file footprint {100 * foot:5.1f} % of the chip over the run", flush=True)
