#!/usr/bin/env python3
"""Where one workgroup of the medium fused path (k_lml_medium, 128 < n <= 4096) spends its cycles, with 512 evaluations
in flight: shader-cycle stamps of workgroup 0 per phase (option diag_stamps).  gpu_medium_phases.py [n ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.lab_context(0)
NAMES = ("build", "diag blocks", "panel solves", "sibling tiles", "trailing tiles", "W step", "Gram + rest")
for n in [int(a) for a in sys.argv[1:]] or [512, 1024, 2048]:
    r = 6
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, r), np.ones((n, 1))], axis=1)
    descs = [gsum_amd.describe_kernel(RBF(0.2 + 0.0001 * i), 1) for i in range(int(os.environ.get("MEDIUM_EVALS", "512")))]
    ctx.set_inputs(X, Z)
    ctx.set_option("medium_min_batch", 1)
    ctx.lml_resident(descs, 1e-10)
    ctx.set_option("diag_stamps", 1)
    ctx.lml_resident(descs, 1e-10)
    v = ctx.diag_stamps_raw()[40:48].astype(float)
    ctx.set_option("diag_stamps", 0)
    tot = v[7]
    print(f"n={n}: workgroup 0 total {tot / 2.4e3:.0f} us (at 2.4 GHz); " + ", ".join(f"{nm} {100 * x / tot:.1f} %" for nm, x in zip(NAMES, v[:7])),
          flush=True)
