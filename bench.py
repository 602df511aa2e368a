#!/usr/bin/env python3
"""bench.py — log-marginal-likelihood throughput of the gsum GP hot path on MI355X.

One "step" = one FULL-RECOMPUTE evaluation, the work of one
``TruncationGP.log_marginal_likelihood(theta, ratio=...)`` call in the reference
(gsum/models.py:958-1039, 1485-1507): RBF kernel-matrix build, jittered Cholesky, forward solve,
Gram / log-det reduction, host scalar algebra.  Workload (BASELINE.json configs[2], SURVEY.md §8d "S3"):
n = 8192 1-D points at dx = 0.5 ell, RBF(ell ~ 0.2), nugget 1e-10, 6 EFT orders, synthetic
coefficients; X and the right-hand sides are resident in HBM before the timed region.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU; every rank evaluates its own K grid points (weak scaling, no data-path
collective) and the fp64 likelihood slices are all-gathered over RCCL inside the timed region.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

# Must precede the first GPU touch of the process (torch included): the HIP runtime reads it when it initialises.  One
# hardware queue per in-flight evaluation instead of 4 shared ones (gsum_amd/_lib.py sets the same default).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X fp64 matrix peak (AMD spec; vector fp64 is the same rate)
HBM_PEAK_GBS = 8000.0


def make_workload(n, r, seed=0):
    import gsum_amd
    X = 0.1 * np.arange(n)[:, None]
    c = np.random.RandomState(seed).randn(n, r)
    y = gsum_amd.partials(c, ratio=0.5, ref=1.0, orders=np.arange(r))
    return X, y


def available_cpus():
    """CPUs this process may really use: affinity mask and cgroup quota, not the host's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(n, r, evals):
    """The CPU oracle (same numpy/scipy/sklearn calls as the reference) on this host's cores, BLAS threads
    = the CPUs this process is allowed to use."""
    from sklearn.gaussian_process.kernels import RBF
    from oracle import gsum_oracle as orc
    threads = available_cpus()
    X, y = make_workload(n, r)
    t = []
    val = None
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:
        limiter = None
    for i in range(evals):
        t0 = time.perf_counter()
        val = orc.trunc_lml(RBF(0.2), np.log([0.2]), X, y, np.arange(r), ratio=0.5, ref=1.0)
        t.append(time.perf_counter() - t0)
    if limiter is not None:
        limiter.restore_original_limits()
    best = min(t)
    return dict(value=1.0 / best, unit="evals/s", cores=int(threads), kind="port",
                sample=f"{evals} full evaluations of oracle.trunc_lml at n={n}, {r} orders (best of {evals}: {best:.2f} s each)",
                lml=float(val))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--orders", type=int, default=6)
    ap.add_argument("--cpu-evals", type=int, default=2, help="CPU-baseline evaluations (0 = skip)")
    ap.add_argument("--slots", type=int, default=0,
                    help="independent evaluations kept in flight per GPU (0 = library default: 20 with 32 hardware "
                         "queues, 3 with the runtime's default 4)")
    ap.add_argument("--backend", default="nccl", help="process-group backend for N > 1 (nccl = RCCL; gloo to rehearse)")
    ap.add_argument("--device", type=int, default=None, help="GPU index override (rehearsal: several ranks on one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = local_rank if args.device is None else args.device
    use_dist = "WORLD_SIZE" in os.environ            # launched by torch.distributed.run (also with one rank)
    if use_dist:
        torch.cuda.set_device(dev)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend)
    elif args.gpus != 1:
        raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if use_dist and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import gsum_amd
    from sklearn.gaussian_process.kernels import RBF
    from gsum_amd.conjugate import lml_from_gram
    from gsum_amd.grid import gather_flat

    n, r, K, W = args.n, args.orders, args.steps, args.warmup
    ctx = gsum_amd.default_context(dev)
    X, y = make_workload(n, r)
    c = gsum_amd.coefficients(y, 0.5, 1.0, np.arange(r))
    Z = np.concatenate([c, np.ones((n, 1))], axis=1)
    jac = float(np.sum(r * np.log(np.abs(np.ones(n))) + np.sum(np.arange(r)) * np.log(np.abs(0.5 * np.ones(n)))))
    ctx.set_inputs(X, Z)                     # X, RHS resident in HBM from here on

    # this rank's grid points: length scales around 0.2 (every point is a distinct evaluation)
    total = world * K
    ells = np.linspace(0.19, 0.21, total) if total > 1 else np.array([0.2])
    mine = ells[rank * K:(rank + 1) * K]
    descs = [gsum_amd.describe_kernel(RBF(float(e)), 1) for e in mine]

    def evaluate(batch):
        """K build + Cholesky + fused solve for every descriptor (the library keeps `slots` independent
        evaluations in flight), then the O(k^2) host algebra per evaluation."""
        G, sld, info = ctx.lml_resident(batch, 1e-10)
        out = np.empty(len(batch))
        for i in range(len(batch)):
            out[i] = -np.inf if info[i] != 0 else lml_from_gram(G[i], sld[i], n, 0.0, 0.0, 1, 1)[0] - jac
        return out

    if args.slots <= 0:                      # the library's own policy (gsum_init)
        nq = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
        args.slots = 20 if nq >= 24 else 14 if nq >= 12 else 10 if nq >= 8 else 3
    ctx.set_option("batch_slots", args.slots)
    # set-up, not a step: the per-slot workspaces (0.5 GB each) are allocated on first use; do that here so that a
    # small --warmup does not leave hipMalloc calls inside the timed region
    evaluate([descs[i % len(descs)] for i in range(args.slots)])
    if W > 0:
        evaluate([descs[i % len(descs)] for i in range(W)])
    if use_dist:
        gather_flat(np.zeros(K), total)          # warm-up of the collective (RCCL sets its rings up lazily)
    PROFILE_EVERY = 4            # HIP events around the bulk launches of every 4th evaluation (every one costs 5 %)
    ctx.set_option("profile_gemm", PROFILE_EVERY)
    ctx.gemm_profile()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vals = evaluate(descs)
    allvals = gather_flat(vals, total) if use_dist else vals       # one all-gather of the fp64 slices (RCCL)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    gemm_ms, gemm_flops, gemm_launches = ctx.gemm_profile()
    ctx.set_option("profile_gemm", 0)
    # single-evaluation stage times (one evaluation alone on the GPU), outside the timed region
    ctx.set_option("batch_slots", 1)
    stage = np.zeros(4)
    for i in range(3):
        ctx.lml_resident([descs[0]], 1e-10)
        tm = ctx.timers()
        stage = np.maximum(stage, 0) if i == 0 else stage
        cur = np.array([tm["build_ms"], tm["potrf_ms"], tm["finalize_ms"], tm["total_ms"]])
        stage = cur if i == 0 else np.minimum(stage, cur)

    # dominant kernel, exclusive: one SYRK launch of the step-0 shape alone on the GPU (device-resident random
    # operands), for the kernel-quality view next to the in-situ numbers
    ctx.bench_gemm_nt(7, n - 256, n - 256, 256, True, n + 16, 2)           # warm-up (first launches read low)
    excl_tflops, excl_us = ctx.bench_gemm_nt(7, n - 256, n - 256, 256, True, n + 16, 4)

    # factor-reuse mode, reported separately and never mixed into `value`: a 64 x 64 (cbar, ratio)
    # grid at ONE kernel costs one K build + Cholesky + forward solve; ratio rescales the Gram matrix order by
    # order, the prior scale (cbar = sd) only enters the O(k^2) host algebra
    reuse = None
    if rank == 0:
        t1 = time.perf_counter()
        Lm = ctx.kernel_matrix_dev(descs[0], X, diag_add=1e-10)
        ctx.potrf(Lm)
        G0, sld0 = ctx.forward_gram(Lm, Z)
        Lm.free()
        grid = np.empty((64, 64))
        for a, q in enumerate(np.linspace(0.3, 0.7, 64)):
            D = np.append((0.5 / q) ** np.arange(r), 1.0)
            Gq = D[:, None] * G0 * D[None, :]
            jq = float(np.sum(np.arange(r)) * np.log(q) * n)
            for b, sd in enumerate(np.linspace(0.5, 2.0, 64)):
                grid[a, b] = lml_from_gram(Gq, sld0, n, 0.0, 0.0, np.inf, sd)[0] - jq
        dt = time.perf_counter() - t1
        reuse = {"grid": "64 x 64 (cbar = sd prior, ratio) at one kernel", "seconds": dt, "evals_per_s": grid.size / dt,
                 "argmax": [int(v) for v in np.unravel_index(np.argmax(grid), grid.shape)], "mode": "factor-reuse"}

    pmc_traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_gemm_pmc.json")) as f:
            pmc_traffic = json.load(f)["derived"]["hbm_traffic_bytes_per_launch"]
    except Exception:
        pass
    if rank == 0:
        potrf_flops = n ** 3 / 3.0
        chol_tflops = potrf_flops / (stage[1] * 1e-3) / 1e12
        # all ranks run the same launches; rank 0's record stands for one GPU
        sampled = (K + PROFILE_EVERY - 1) // PROFILE_EVERY          # evaluations 0, 4, 8, ... of the timed K
        chip_tflops = gemm_flops * (K / sampled) / elapsed / 1e12     # every evaluation issues the same launches
        out = {
            "metric": "lml_evals_per_sec", "value": total / elapsed, "unit": "evals/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"full-recompute lml eval: n={n} 1-D RBF(ell~0.2) dx=0.5ell, nugget 1e-10, "
                                   f"{r} orders (BASELINE configs[2], S3)", "n": n, "orders": r,
                       "evals_per_gpu": K, "mode": "full-recompute", "evals_in_flight_per_gpu": args.slots},
            "single_eval_stage_ms": {"kernel_build": stage[0], "cholesky_fused_solve": stage[1],
                                     "finalize_d2h": stage[2], "gpu_total": stage[3]},
            "cholesky": {"single_eval_gflops": chol_tflops * 1e3,
                         "single_eval_frac_of_fp64_mfma_peak": chol_tflops / FP64_MFMA_PEAK_TFLOPS,
                         "pipelined_gflops_per_gpu": potrf_flops * K / elapsed / 1e9,
                         "pipelined_frac_of_fp64_mfma_peak": potrf_flops * K / elapsed / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                         "flops": "n^3/3", "single_eval_ms": stage[1]},
            "kernel_build": {"gbps_algorithmic_8n2": 8.0 * n * n / (stage[0] * 1e-3) / 1e9,
                             "gbps_bytes_written": (4.0 * n * n + 4.0 * n * 128) / (stage[0] * 1e-3) / 1e9,
                             "bytes_written": "lower-triangle tiles only (4n^2 + 4n*128)"},
            # dominant kernel: the 128x128-tile fp64-MFMA SYRK of the trailing update.  `achieved` = algorithmic
            # flops of ALL its launches in the timed region / wall time of the timed region, i.e. what this
            # kernel delivers on the chip while `evals_in_flight` evaluations share it; per-launch averages
            # (HIP events on the launch stream, what rocprofv3 --stats reports) and the exclusive rate follow.
            "roofline": {"kernel": "k_gemm_ld3 (128x64-tile, 8-wave fp64 MFMA SYRK with LDS-direct operand staging, 3 workgroups "
                                   "per CU, K=256/512, trailing update)",
                         "bound": "mfma", "achieved": chip_tflops, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": chip_tflops / FP64_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic,
                         "traffic_source": "HBM bytes of one M=8192, K=256 launch of this kernel (FETCH_SIZE x 2 + WRITE_SIZE) from "
                                           "the committed rocprofv3 --pmc passes, profiles/r01_gemm_pmc.json / .md; PMC cannot be "
                                           "collected inside bench.py; algorithmic bytes of that launch: 5.61e8",
                         "launches": gemm_launches, "launches_sampled": f"every {PROFILE_EVERY}th evaluation of the timed region",
                         "avg_launch_us": gemm_ms * 1e3 / max(1, gemm_launches),
                         "avg_flops_per_launch": gemm_flops / max(1, gemm_launches),
                         "avg_concurrent_launches": gemm_ms * 1e-3 * (K / sampled) / elapsed,
                         "per_launch_tflops_shared": gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0,
                         "exclusive_tflops": excl_tflops, "exclusive_frac": excl_tflops / FP64_MFMA_PEAK_TFLOPS,
                         "flops_per_launch": "algorithmic flops of each launch: lower-triangular SYRK M(M+1)K (K = 256, or 512 for "
                                             "the lazily updated far region), lower trapezoid for the near-column updates"},
            "factor_reuse": reuse,
            "lml_sample": float(allvals[0]),
        }
        if world == 1 and args.cpu_evals > 0:
            out["cpu_baseline"] = cpu_baseline(n, r, args.cpu_evals)
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
