import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import gsum_amd
from sklearn.gaussian_process.kernels import RBF
mode = sys.argv[1] if len(sys.argv) > 1 else "nccl"
ctx = gsum_amd.default_context(0)
n, r = 8192, 6
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, r), np.ones((n, 1))], axis=1)
ctx.set_inputs(X, Z)
descs = [gsum_amd.describe_kernel(RBF(float(e)), 1) for e in np.linspace(0.19, 0.21, 40)]
def run(label):
    ctx.lml_resident(descs[:20], 1e-10)
    t0 = time.perf_counter(); ctx.lml_resident(descs, 1e-10); dt = time.perf_counter() - t0
    ctx.set_option("batch_slots", 1)
    ts = []
    for _ in range(3):
        ctx.lml_resident(descs[:1], 1e-10); ts.append(ctx.timers()["potrf_ms"])
    ctx.set_option("batch_slots", 20)
    ex = ctx.bench_gemm_nt(7, n - 256, n - 256, 256, True, n + 16, 4)
    ctx.set_option("batch_slots", 3)
    t0 = time.perf_counter(); ctx.lml_resident(descs[:12], 1e-10); d3 = time.perf_counter() - t0
    ctx.set_option("batch_slots", 20)
    print(f"{label:46s} {dt / 40 * 1e3:.3f} ms/eval (20 slots)  {d3 / 12 * 1e3:.3f} (3 slots)  single potrf {min(ts):.3f} ms  excl gemm {ex[0]:.1f} TF/s  probe {ctx.queue_probe()['concurrency']}", flush=True)
run("before any process group")
if mode != "none":
    torch.cuda.set_device(0)
    dist.init_process_group(mode, rank=0, world_size=1, **({"device_id": torch.device("cuda", 0)} if mode == "nccl" else {}))
    run("after init_process_group (communicator lazy)")
    t = torch.ones(4, device="cuda" if mode == "nccl" else "cpu"); dist.all_reduce(t); torch.cuda.synchronize()
    run("after the first collective")
    run("again")
    dist.destroy_process_group()
    run("after destroy_process_group")
