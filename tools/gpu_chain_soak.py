#!/usr/bin/env python3
"""Soak of the persistent-chain schedule under UNEVEN load: a second context on the same card runs rectangular MFMA launches of
varying size from another thread while the chain schedule factorises; every evaluation is compared bit for bit (G, sum log L_ii,
info) with the host-enqueued schedule's result on the same inputs.  A stale word anywhere in a hand-off changes the result, so
"identical" over many perturbed runs is the check MI355X_MICROARCH.md asks of every inter-workgroup hand-off.
Usage: gpu_chain_soak.py [seconds per size]"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from gsum_amd._lib import HipContext  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
ctx = gsum_amd.lab_context(0)
ctx.set_option("batch_slots", 1)
noise = HipContext(0)
stop = threading.Event()
launched = [0]


def disturb():
    rng = np.random.RandomState(7)
    while not stop.is_set():
        m = int(rng.choice([256, 1024, 2048, 4096]))
        noise.bench_gemm_nt(7, m, m, 256, tri=False, reps=int(rng.randint(1, 4)))
        launched[0] += 1
        time.sleep(float(rng.rand()) * 2e-3)


out, bad = [], 0
for n, load in ((2048, False), (2048, True), (4096, True), (8192, False), (8192, True), (6000, True)):
    rng = np.random.RandomState(n)
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([rng.randn(n, 6), np.ones((n, 1))], axis=1)
    desc = gsum_amd.describe_kernel(RBF(0.2), 1)
    ctx.set_inputs(X, Z)
    ctx.set_option("chain_persist", 0)
    G0, s0, i0 = ctx.lml_resident([desc], 1e-10)
    G0, s0, i0 = G0[0].copy(), float(s0[0]), int(i0[0])
    ctx.set_option("release_scratch", 1)
    ctx.set_option("chain_persist", 1)
    th = None
    if load:
        stop.clear()
        launched[0] = 0
        th = threading.Thread(target=disturb, daemon=True)
        th.start()
    runs = same = 0
    ts = []
    t_end = time.time() + budget
    while time.time() < t_end:
        if ctx.get_option("chain_persist") != 1:        # a give-up switches the schedule off for the context: count it, switch back on
            ctx.set_option("chain_persist", 1)
        G, s, i = ctx.lml_resident([desc], 1e-10)
        ts.append(ctx.timers()["potrf_ms"])
        runs += 1
        same += bool(np.array_equal(G[0], G0) and float(s[0]) == s0 and int(i[0]) == i0)
    if th is not None:
        stop.set()
        th.join()
    rec = dict(n=n, loaded=load, runs=runs, identical=same, aborts=ctx.get_option("chain_aborts"), noise_launches=launched[0] if load else 0,
               potrf_ms_min=min(ts), potrf_ms_median=float(np.median(ts)), potrf_ms_max=max(ts))
    bad += same != runs
    out.append(rec)
    print(json.dumps(rec), flush=True)

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "chain_soak.json"), "w") as f:
    json.dump(out, f, indent=1)
print("SOAK", "FAILED" if bad else "ok", flush=True)
sys.exit(1 if bad else 0)
