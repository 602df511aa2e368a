#!/usr/bin/env python3
"""One factorisation on the persistent-chain schedule with its far updates at K = 256 every step (chain_lazy = 0), K = 512 every other step with the next two panels'
columns near (1) or only the next panel's (2): n = 8192, 12288, 16384, same process, interleaved; results must be bit-identical."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

ctx = gsum_amd.lab_context(0)
ctx.set_option("batch_slots", 1)
for n in [int(a) for a in sys.argv[1:]] or (8192, 12288, 16384):
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
    ctx.set_inputs(X, Z)
    d = gsum_amd.describe_kernel(RBF(0.2), 1)
    out, ref = [], None
    for rnd in range(2):
        for lazy in (0, 1, 2):
            ctx.set_option("chain_lazy", lazy)
            ctx.lml_resident([d], 1e-10)
            ts = []
            for _ in range(4):
                G, s, i = ctx.lml_resident([d], 1e-10)
                ts.append(ctx.timers()["potrf_ms"])
            key = (float(s[0]).hex(), float(G[0, 1, 2]).hex(), int(i[0]))
            ref = ref or key
            out.append(f"lazy={lazy}: {min(ts):.3f} ms{'' if key == ref else ' DIFFERENT'}")
    print(f"n={n}: " + "  ".join(out) + f"  time-outs {ctx.get_option('chain_aborts')}", flush=True)
ctx.set_option("chain_lazy", -1)
