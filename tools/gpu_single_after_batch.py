#!/usr/bin/env python3
"""Does ONE evaluation keep its persistent-chain schedule after a batch call has created the groups' streams?
   Prints potrf ms of single evaluations on a fresh context, after a batch of 20, and on a second context."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
if len(sys.argv) > 2 and sys.argv[2] == "torch":          # the way bench.py runs: torch has initialised the device first
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
    print("torch initialised", flush=True)
rng = np.random.default_rng(0)
X = (0.1 * np.arange(n))[:, None]
Z = np.concatenate([rng.standard_normal((n, 6)), np.ones((n, 1))], axis=1)
descs = [gsum_amd.describe_kernel(RBF(0.2 + 1e-4 * i), 1) for i in range(20)]


def singles(ctx, tag):
    ms = []
    for _ in range(5):
        ctx.lml_resident([descs[0]], 1e-10)
        ms.append(round(ctx.timers()["potrf_ms"], 3))
    out = {"what": tag, "potrf_ms": ms, "chain_probe": ctx.get_option("chain_probe"), "chain_aborts": ctx.get_option("chain_aborts"),
           "chain_persist": ctx.get_option("chain_persist")}
    print(json.dumps(out), flush=True)


ctx = gsum_amd.HipContext(0)
ctx.set_inputs(X, Z)
singles(ctx, "fresh context")
ctx.lml_resident(descs, 1e-10)
ctx.lml_resident(descs, 1e-10)
singles(ctx, "after two batch calls of 20")
ctx.set_option("profile_gemm", 1)
ctx.kernel_profile()
ctx.lml_resident(descs, 1e-10)
ctx.kernel_profile()
ctx.set_option("profile_gemm", 0)
singles(ctx, "after a profiled batch call")
ctx2 = gsum_amd.HipContext(0)
ctx2.set_inputs(X, Z)
ctx2.lml_resident(descs, 1e-10)
singles(ctx2, "second context, batch first")
