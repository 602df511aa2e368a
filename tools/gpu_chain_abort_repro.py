#!/usr/bin/env python3
"""Find the call sequence after which ONE evaluation's persistent chain times out (bench.py saw chain_aborts = 1).
   usage: gpu_chain_abort_repro.py VARIANT     (each variant in a process of its own)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "bench"
n = 8192
rng = np.random.default_rng(0)
X = (0.1 * np.arange(n))[:, None]
Z = np.concatenate([rng.standard_normal((n, 6)), np.ones((n, 1))], axis=1)
descs = [gsum_amd.describe_kernel(RBF(0.19 + 1e-3 * i), 1) for i in range(20)]
if "torch" in variant:
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
if "nccl" in variant:          # a rank under torch.distributed.run: the RCCL communicator (and its streams) exists before the context
    import torch.distributed as dist
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(8, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
ctx = gsum_amd.HipContext(0)
ctx.set_inputs(X, Z)


def state(tag, t0):
    print(json.dumps({"variant": variant, "after": tag, "wall_ms": round((time.perf_counter() - t0) * 1e3, 2), "potrf_ms": round(ctx.timers()["potrf_ms"], 3),
                      "aborts": ctx.get_option("chain_aborts"), "persist": ctx.get_option("chain_persist"), "probe": ctx.get_option("chain_probe")}), flush=True)


def single(tag, d=None):
    t0 = time.perf_counter()
    ctx.lml_resident([d or descs[0]], 1e-10)
    state(tag, t0)


def batch(k):
    ctx.lml_resident([descs[i % 20] for i in range(k)], 1e-10)


if variant.startswith("bench"):
    batch(20); batch(3); 
    for _ in range(3):
        batch(20)
    single("20,3,20x3")
    single("again")
elif variant == "fresh":
    single("nothing"); single("again"); batch(20); single("then 20")
elif variant == "torch_nccl_b20":
    batch(20); single("nccl, 20"); single("again"); batch(20); single("20 again")
elif variant == "b20_b3":
    batch(20); batch(3); single("20,3"); single("again")
elif variant == "b3":
    batch(3); single("3"); single("again")
elif variant == "b20":
    batch(20); single("20"); single("again")
elif variant == "b20_new_desc":
    batch(20); single("20, new descriptor", gsum_amd.describe_kernel(RBF(0.19), 1)); single("again")
elif variant == "many":
    for _ in range(12):
        batch(20)
    single("12 x 20"); single("again")
