#!/usr/bin/env python3
"""Deep-grouped head of the persistent-chain schedule (round 5, option chain_deep): bit-identity of G / sum log L_ii / info and of the
factor itself against the per-step schedule (chain_deep = 0), and the latency of ONE factorisation alone, per order and depth.
Usage: gpu_chain_deep.py [reps] [orders...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
orders = [int(v) for v in sys.argv[2:]] or [4096, 6000, 8192, 12288, 16384]
ctx = gsum_amd.lab_context(0)
bad = 0


def evaluate(desc):
    G, sld, info = ctx.lml_resident([desc], 1e-10)
    return G[0].copy(), float(sld[0]), int(info[0]), ctx.timers()["potrf_ms"]


for n in orders:
    rng = np.random.RandomState(n)
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([rng.randn(n, 6), np.ones((n, 1))], axis=1)
    desc = gsum_amd.describe_kernel(RBF(0.2), 1)
    ctx.set_inputs(X, Z)
    ctx.set_option("chain_persist", 1)
    ctx.set_option("chain_deep", 0)
    ref = evaluate(desc)
    t_ref = sorted(evaluate(desc)[3] for _ in range(reps))
    print(json.dumps(dict(n=n, deep=0, potrf_ms_min=t_ref[0], potrf_ms_median=t_ref[len(t_ref) // 2], info=ref[2])), flush=True)
    for depth, rows in ((4, 2560), (3, 2560), (2, 2560), (4, 1536), (4, 4096), (6, 2560), (8, 2560)):
        ctx.set_option("chain_deep", 1)
        ctx.set_option("chain_depth", depth)
        ctx.set_option("chain_deep_rows", rows)
        same, ts = 0, []
        for _ in range(reps):
            G, sld, info, ms = evaluate(desc)
            same += bool(np.array_equal(G, ref[0]) and sld == ref[1] and info == ref[2])
            ts.append(ms)
        ts.sort()
        bad += same != reps
        print(json.dumps(dict(n=n, deep=1, depth=depth, deep_rows=rows, identical=same, reps=reps, potrf_ms_min=ts[0],
                              potrf_ms_median=ts[len(ts) // 2], aborts=ctx.get_option("chain_aborts"),
                              persist_now=ctx.get_option("chain_persist"))), flush=True)
        if ctx.get_option("chain_persist") == 0:
            print("chain schedule switched itself off (time-out): stopping", flush=True)
            sys.exit(2)
    ctx.set_option("chain_depth", 4)
    ctx.set_option("chain_deep_rows", 2560)
    ctx.set_option("release_scratch", 1)

# the factor itself (operator level), n = 8192: deep against per-step
n = 8192
X = 0.1 * np.arange(n)[:, None]
desc = gsum_amd.describe_kernel(RBF(0.2), 1)
Ls = []
for deep in (0, 1):
    ctx.set_option("chain_deep", deep)
    K = ctx.kernel_matrix_dev(desc, X, diag_add=1e-10)
    info = ctx.potrf(K)
    Ls.append((info, K.to_host()))
    K.free()
same = bool(np.array_equal(Ls[0][1], Ls[1][1]))
print("factor identical (n = 8192):", same, "info", Ls[0][0], Ls[1][0], flush=True)
bad += not same
print("MISMATCHES" if bad else "all identical", flush=True)
sys.exit(1 if bad else 0)
