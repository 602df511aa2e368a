#!/usr/bin/env python3
"""Round-2 kernel A/B on the GPU box: diagonal-block kernel (diag_algo 1 vs 2) and kernel build (build_algo 1 vs 2):
bit / tolerance checks between the two, in-kernel stamps, stage timings, single-evaluation and batch rates.
Writes gpurun_out/r2_kernels.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, Matern, WhiteKernel, ConstantKernel as C  # noqa: E402

out = {}
ctx = gsum_amd.lab_context(0)

# ---- kernel build: equality of the two kernels, then timing
for name, kern, d in (("rbf1d", RBF(0.2), 1), ("m52_2d", C(1.3) * Matern([0.7, 1.3], nu=2.5) + WhiteKernel(1e-3), 2),
                      ("m32_3d", Matern(0.9, nu=1.5) + C(0.2), 3), ("m12_1d", Matern(0.5, nu=0.5), 1)):
    rng = np.random.RandomState(3)
    for n in (333, 1000):
        X = rng.rand(n, d) * 7
        desc = gsum_amd.describe_kernel(kern, d)
        res = {}
        for algo in (1, 2):
            ctx.set_option("build_algo", algo)
            res[algo] = (ctx.kernel_matrix(desc, X, diag_add=1e-7), ctx.kernel_matrix(desc, X, X[:77] + 0.01))
        same = all(np.array_equal(res[1][i], res[2][i]) for i in (0, 1))
        ref = kern(X)
        ref[np.diag_indices_from(ref)] += 1e-7
        print("build", name, n, "algo1 == algo2:", same, "== sklearn:", np.array_equal(res[2][0], ref),
              np.array_equal(res[2][1], kern(X, X[:77] + 0.01)), flush=True)
        out[f"build_equal_{name}_{n}"] = bool(same)

n, r = 8192, 6
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, r), np.ones((n, 1))], axis=1)
desc = gsum_amd.describe_kernel(RBF(0.2), 1)
ctx.set_inputs(X, Z)
ref = None
for balgo in (1, 2):
    for dalgo in (1, 2):
        ctx.set_option("build_algo", balgo)
        ctx.set_option("diag_algo", dalgo)
        ctx.set_option("batch_slots", 1)
        ts = []
        for rep in range(6):
            G, sld, info = ctx.lml_resident([desc], 1e-10)
            ts.append(ctx.timers())
        best = {k: min(t[k] for t in ts) for k in ts[0]}
        key = f"n8192_build{balgo}_diag{dalgo}"
        out[key] = best
        if ref is None:
            ref = (G.copy(), sld.copy())
        relG = float(np.max(np.abs(G - ref[0]) / np.abs(ref[0]).max()))
        print(key, {k: round(v, 4) for k, v in best.items()}, "info", int(info[0]), "relG vs first", relG, "dsld", float(sld[0] - ref[1][0]), flush=True)
        out[key]["relG_vs_first"] = relG
ctx.set_option("build_algo", 2)

# ---- diagonal kernel stamps (one 128-block inside an n = 2048 evaluation)
n2 = 2048
X2 = 0.1 * np.arange(n2)[:, None]
Z2 = np.concatenate([np.random.RandomState(0).randn(n2, 4), np.ones((n2, 1))], axis=1)
ctx.set_inputs(X2, Z2)
ctx.set_option("diag_stamps", 1)
ctx.set_option("medium_path", 0)
for dalgo in (1, 2):
    ctx.set_option("diag_algo", dalgo)
    for rep in range(3):
        ctx.lml_resident([desc], 1e-10)
    st = ctx.diag_stamps()
    out[f"diag_stamps_algo{dalgo}"] = st
    print("diag stamps algo", dalgo, st, flush=True)
    if dalgo == 2:
        raw = ctx.diag_stamps_raw()
        t0 = raw[7]
        print("  barrier stamps rel. start:", [int(v - t0) for v in raw[8:24]], "potf2 cycles:", [int(v) for v in raw[24:32]], flush=True)
ctx.set_option("diag_stamps", 0)
ctx.set_option("medium_path", 1)

# ---- potrf against LAPACK with both kernels
from scipy.linalg.lapack import dpotrf  # noqa: E402
for nn in (1, 16, 17, 128, 129, 200, 1000):
    rng = np.random.RandomState(nn)
    Xr = np.sort(rng.rand(nn, 1), axis=0) * nn * 0.15
    K = RBF(0.3)(Xr) + 1e-8 * np.eye(nn)
    Lref = np.linalg.cholesky(K)
    for dalgo in (1, 2):
        ctx.set_option("diag_algo", dalgo)
        M = ctx.upload(K)
        info = ctx.potrf(M)
        L = M.to_host()
        M.free()
        err = float(np.max(np.abs(L - Lref)) / np.max(np.abs(Lref)))
        res = float(np.max(np.abs(L @ L.T - K)))
        print("potrf", nn, "algo", dalgo, "info", info, "max|L - Lref|", err, "max|LL^T - K|", res, flush=True)
        out[f"potrf_{nn}_algo{dalgo}"] = dict(info=info, err=err, res=res)

# ---- batch throughput with both diagonal kernels
ctx.set_inputs(X, Z)
descs = [gsum_amd.describe_kernel(RBF(float(e)), 1) for e in np.linspace(0.19, 0.21, 40)]
for dalgo in (1, 2, 1, 2):
    ctx.set_option("diag_algo", dalgo)
    ctx.set_option("batch_slots", 20)
    ctx.lml_resident(descs[:20], 1e-10)
    t0 = time.perf_counter()
    ctx.lml_resident(descs, 1e-10)
    dt = time.perf_counter() - t0
    print("batch 40 evals, diag algo", dalgo, "ms/eval", dt / 40 * 1e3, flush=True)
    out.setdefault(f"batch_ms_per_eval_algo{dalgo}", []).append(dt / 40 * 1e3)
print("queue probe", ctx.queue_probe(), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r2_kernels.json"), "w") as f:
    json.dump(out, f, indent=1, default=float)
