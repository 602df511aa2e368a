#!/usr/bin/env python3
"""From how many evaluations on the fused one-workgroup-per-evaluation path (k_lml_medium) beats the pipelined multi-kernel path: wall time of a
call of c evaluations on each (options medium_path / medium_min_batch), n = 256 ... 4096."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.lab_context(0)

for n in (256, 512, 1024, 1536, 2048, 3072, 4096):
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, 4), np.ones((n, 1))], axis=1)
    ctx.set_inputs(X, Z)
    row, cross = [], None
    for c in (2, 4, 8, 16, 32, 64, 96, 128, 192, 256, 384, 512):
        descs = ctx.desc_array([gsum_amd.describe_kernel(RBF(float(e)), 1) for e in np.linspace(0.15, 0.25, c)])
        t = {}
        for name, opts in (("fused", (("medium_path", 1), ("medium_min_batch", 1))), ("pipelined", (("medium_path", 0),))):
            for k, v in opts:
                ctx.set_option(k, v)
            ctx.lml_resident(descs, 1e-10)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                ctx.lml_resident(descs, 1e-10)
                ts.append((time.perf_counter() - t0) * 1e3)
            t[name] = min(ts)
        row.append(f"{c}: {t['fused']:.2f}/{t['pipelined']:.2f}")
        if cross is None and t["fused"] < t["pipelined"]:
            cross = c
    rule = max(4, int(n ** 1.55 / 2000.0))
    print(f"n={n:5d}: fused wins from c = {cross} (rule in the library: {rule});  ms fused/pipelined  " + "  ".join(row), flush=True)
ctx.set_option("medium_path", 1)
ctx.set_option("medium_min_batch", 0)
