#!/usr/bin/env python3
"""Grouped batch schedule (gs_lml_wave, round 4) against single evaluations (rounds 1-3's one-stream-per-evaluation batch, which
the first version of this script compared with -- bit-identical, 215 evals/s on default queues -- is gone):
bit-identity of G / sum log L_ii / info on mixed batches (one non-positive-definite member), then throughput at
n = 8192 for several group layouts.  GPU_MAX_HW_QUEUES is deliberately NOT set by this script.

    python tools/gpu_wave_check.py [--quick] [--n 8192] [--evals 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def workload(n, r, seed=0):
    import gsum_amd
    X = 0.1 * np.arange(n)[:, None]
    c = np.random.RandomState(seed).randn(n, r)
    y = gsum_amd.partials(c, ratio=0.5, ref=1.0, orders=np.arange(r))
    cc = gsum_amd.coefficients(y, 0.5, 1.0, np.arange(r))
    return X, np.concatenate([cc, np.ones((n, 1))], axis=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--evals", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--skip-identity", action="store_true")
    args = ap.parse_args()
    import gsum_amd
    from sklearn.gaussian_process.kernels import RBF, Matern, WhiteKernel
    ctx = gsum_amd.lab_context(0)
    print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
    out = {"identity": [], "timing": []}

    if not args.skip_identity:
        ctx.set_option("medium_path", 0)
        sizes = [(300, 4), (1000, 3), (2048, 4), (4352, 6)] + ([] if args.quick else [(8192, 6)])
        for n, r in sizes:
            X, Z = workload(n, r)
            ctx.set_inputs(X, Z)
            ells = np.linspace(0.17, 0.23, 7)
            descs = [gsum_amd.describe_kernel(RBF(float(e)), 1) for e in ells]
            descs.append(gsum_amd.describe_kernel(Matern(0.2, nu=2.5) + WhiteKernel(1e-6, noise_level_bounds="fixed"), 1))
            descs.append(gsum_amd.describe_kernel(RBF(30.0), 1))                  # numerically singular: info > 0
            nug = 1e-10
            ctx.set_option("wave_min", 1000)              # the reference: one evaluation after the other, single-factorisation schedule
            G0, s0, i0 = ctx.lml_resident(descs, nug)
            ctx.set_option("wave_min", 3)
            res = {"n": n, "evals": len(descs), "info": i0.tolist()}
            for (g, b) in ((2, 10), (3, 2), (1, 4), (4, 1)):
                ctx.set_option("wave_groups", g)
                ctx.set_option("wave_size", b)
                G1, s1, i1 = ctx.lml_resident(descs, nug)
                ok_rows = i0 == 0
                same = bool(np.array_equal(i0, i1) and np.array_equal(G0[ok_rows], G1[ok_rows]) and np.array_equal(s0[ok_rows], s1[ok_rows]))
                res[f"g{g}b{b}"] = same
                if not same:
                    res[f"g{g}b{b}_maxdiff"] = float(np.max(np.abs(G0[ok_rows] - G1[ok_rows])))
            print(json.dumps(res), flush=True)
            out["identity"].append(res)
        ctx.set_option("medium_path", 1)
        ctx.set_option("release_scratch", 1)

    n = args.n
    X, Z = workload(n, 6)
    ctx.set_inputs(X, Z)

    def rate(K, reps):
        ells = np.linspace(0.19, 0.21, K)
        descs = ctx.desc_array([gsum_amd.describe_kernel(RBF(float(e)), 1) for e in ells])
        ctx.lml_resident(descs, 1e-10)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            ctx.lml_resident(descs, 1e-10)
            ts.append(time.perf_counter() - t0)
        return K / float(np.median(ts)), K / min(ts)

    K = args.evals
    layouts = [(2, 10, 1), (2, 10, 0), (3, 7, 1), (4, 5, 1), (1, 20, 1), (2, 12, 1)] if not args.quick else [(2, 10, 1), (3, 7, 1)]
    for g, b, near in layouts:
        ctx.set_option("wave_groups", g)
        ctx.set_option("wave_size", b)
        ctx.set_option("wave_near_on_chain", near)
        med, best = rate(K, args.reps)
        rec = {"mode": "wave", "groups": g, "size": b, "near_on_chain": near, "K": K, "evals_per_s_median": med, "best": best}
        print(json.dumps(rec), flush=True)
        out["timing"].append(rec)
    # deeper grouping of the trailing updates (K = 256 x depth for the far region)
    ctx.set_option("wave_groups", 2)
    ctx.set_option("wave_size", 10)
    ctx.set_option("wave_near_on_chain", 1)
    ells = np.linspace(0.19, 0.21, K)
    dd = ctx.desc_array([gsum_amd.describe_kernel(RBF(float(e)), 1) for e in ells])
    ref = ctx.lml_resident(dd, 1e-10)
    for depth, rows in ((2, 3072), (3, 3072), (4, 3072), (4, 1024), (4, 5120), (6, 4096), (8, 4096), (1, 0), (2, 3072)):
        ctx.set_option("wave_depth", depth)
        ctx.set_option("wave_deep_rows", rows)
        got = ctx.lml_resident(dd, 1e-10)
        same = all(np.array_equal(a, b) for a, b in zip(ref, got))
        med, best = rate(K, args.reps)
        rec = {"mode": "wave", "depth": depth, "deep_rows": rows, "K": K, "bit_identical": same, "evals_per_s_median": med, "best": best}
        print(json.dumps(rec), flush=True)
        out["timing"].append(rec)
    ctx.set_option("wave_depth", 2)
    ctx.set_option("wave_deep_rows", 3072)
    # per-class HIP-event times of one profiled call in the default layout
    ctx.set_option("wave_groups", 2)
    ctx.set_option("wave_size", 10)
    ctx.set_option("wave_near_on_chain", 1)
    ells = np.linspace(0.19, 0.21, K)
    descs = ctx.desc_array([gsum_amd.describe_kernel(RBF(float(e)), 1) for e in ells])
    ctx.set_option("profile_gemm", 1)
    ctx.kernel_profile()
    t0 = time.perf_counter()
    ctx.lml_resident(descs, 1e-10)
    wall = time.perf_counter() - t0
    prof = ctx.kernel_profile()
    ctx.set_option("profile_gemm", 0)
    rec = {"profiled_call_ms": wall * 1e3, "classes": prof,
           "bulk_tflops_per_launch_time": prof["bulk_update"]["flops"] / (prof["bulk_update"]["ms"] * 1e-3) / 1e12}
    print(json.dumps(rec), flush=True)
    out["profile"] = rec
    if not args.quick:
        # long calls: several rounds per group, groups out of phase by wave_shift outer steps
        for g, b, shift, KK in ((2, 10, -1, 80), (2, 10, 0, 80), (2, 10, 4, 80), (3, 7, -1, 84), (3, 7, 0, 84), (2, 10, -1, 200)):
            ctx.set_option("wave_groups", g)
            ctx.set_option("wave_size", b)
            ctx.set_option("wave_shift", shift)
            med, best = rate(KK, 3)
            rec = {"mode": "wave", "groups": g, "size": b, "shift": shift, "K": KK, "evals_per_s_median": med, "best": best}
            print(json.dumps(rec), flush=True)
            out["timing"].append(rec)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "wave_check.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
