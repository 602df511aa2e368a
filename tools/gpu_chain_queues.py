#!/usr/bin/env python3
"""The chain schedule when the process has only the HIP runtime's default hardware queues (GPU_MAX_HW_QUEUES unset = 4), or few: does
the stream-concurrency probe keep the schedule off where its streams would share a queue, does any party time out, and what does one
factorisation cost.  One child process per setting (the variable is read when the runtime starts).  Usage: gpu_chain_queues.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import numpy as np
import gsum_amd
from sklearn.gaussian_process.kernels import RBF
ctx = gsum_amd.lab_context(0)
ctx.set_option("batch_slots", 1)
out = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
for n in (2048, 8192):
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
    desc = gsum_amd.describe_kernel(RBF(0.2), 1)
    ctx.set_inputs(X, Z)
    ctx.set_option("chain_persist", 0)
    G0, s0, i0 = ctx.lml_resident([desc], 1e-10)
    t_host = min((ctx.lml_resident([desc], 1e-10), ctx.timers()["potrf_ms"])[1] for _ in range(4))
    ctx.set_option("release_scratch", 1)
    ctx.set_option("chain_persist", -1)
    ts, same, wall = [], 0, []
    for _ in range(6):
        t0 = time.perf_counter()
        G, s, i = ctx.lml_resident([desc], 1e-10)
        wall.append((time.perf_counter() - t0) * 1e3)
        ts.append(ctx.timers()["potrf_ms"])
        same += bool(np.array_equal(G, G0) and s[0] == s0[0] and i[0] == i0[0])
    out[str(n)] = dict(potrf_ms_host_schedule=round(t_host, 3), potrf_ms_auto=[round(t, 3) for t in ts], wall_ms_auto_max=round(max(wall), 1),
                       identical=same, chain_probe=ctx.get_option("chain_probe"), chain_persist_after=ctx.get_option("chain_persist"),
                       chain_aborts=ctx.get_option("chain_aborts"))
print("RESULT " + json.dumps(out))
''' % ROOT

for q in (None, "4", "8", "32"):
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)
    if q is not None:
        env["GPU_MAX_HW_QUEUES"] = q
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=150)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    print(line[0][7:] if line else f"queues={q}: rc={r.returncode} {r.stderr[-600:]}", flush=True)
