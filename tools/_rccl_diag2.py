import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import gsum_amd
ctx = gsum_amd.default_context(0)
n = 8192
def run(label):
    ex = ctx.bench_gemm_nt(7, n - 256, n - 256, 256, True, n + 16, 6)
    pr = ctx.probe_mfma_f64(20000, 2, 8)
    hb = ctx.probe_hbm_write(1 << 30)
    print(f"{label:46s} excl gemm {ex[0]:.1f} TF/s ({ex[1]:.0f} us)  mfma probe {pr['tflops']:.1f} TF/s clock {pr['clock_ghz']:.3f} GHz  hbm write {hb:.0f} GB/s", flush=True)
for i in range(2): run("before")
mode = sys.argv[1]
if mode == "nccl":
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
elif mode == "torchonly":
    torch.cuda.set_device(0); x = torch.ones(1 << 20, device="cuda"); torch.cuda.synchronize()
elif mode == "rsmi":
    import subprocess; subprocess.run(["rocm-smi", "--showclocks"], capture_output=True)
for i in range(3): run(f"after {mode}")
time.sleep(5)
run("5 s later")
