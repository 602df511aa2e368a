#!/usr/bin/env python3
"""One value + gradient evaluation alone (gsum_lml_grad): the U = L^-T sweep's launches enqueued between the factorisation's outer steps
(grad_interleave = 1, gs_potrf_chain's step hook) against enqueued behind the whole factorisation (0).  Same kernels on the same data:
results must be bit-identical."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from gsum_amd.kernels import describe_kernel, describe_gradient  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C  # noqa: E402

ctx = gsum_amd.lab_context(0)
sizes = [int(a) for a in sys.argv[1:]] or [1500, 2048, 4096, 8192, 12288]
for n in sizes:
    r = 6
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, r), np.ones((n, 1))], axis=1)
    kern = C(1.0) * RBF(0.2) + WhiteKernel(1e-8)
    desc, prm = describe_kernel(kern, 1), describe_gradient(kern, 1)
    res = {}
    for mode in (0, 1, 0, 1):
        # 0: the round-4 form; 1: sweep launches interleaved + contractions split (Q beside the R^-1 product, traces from stored dR triangles)
        ctx.set_option("grad_interleave", mode)
        ctx.set_option("grad_split", mode)
        ctx.set_option("grad_lazy_chain", mode)
        out = ctx.lml_grad(desc, prm, X, Z, 1e-10)
        reps = 6 if n <= 8192 else 3
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = ctx.lml_grad(desc, prm, X, Z, 1e-10)
            ts.append(time.perf_counter() - t0)
        res.setdefault(mode, []).append((min(ts) * 1e3, float(np.median(ts)) * 1e3, out))
    same = all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(res[0][0][2], res[1][0][2]))
    print(f"n={n}: round-4 form {res[0][0][0]:.2f} / {res[0][1][0]:.2f} ms (best of two rounds' minima; medians {res[0][0][1]:.2f} / {res[0][1][1]:.2f}), "
          f"interleaved + split {res[1][0][0]:.2f} / {res[1][1][0]:.2f} ms (medians {res[1][0][1]:.2f} / {res[1][1][1]:.2f}), identical {same}, "
          f"chain time-outs {ctx.get_option('chain_aborts') if hasattr(ctx, 'get_option') else '?'}", flush=True)
    ctx.set_option("release_scratch", 1)
ctx.set_option("grad_interleave", 1)
ctx.set_option("grad_split", 1)
ctx.set_option("grad_lazy_chain", 1)
