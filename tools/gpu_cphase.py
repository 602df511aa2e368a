import os, sys
sys.path.insert(0, "/root/repo")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, gsum_amd
ctx = gsum_amd.default_context(0)
ctx.bench_gemm_nt(7, 7936, 7936, 256, True, 8208)
for rnd in range(2):
    for beta, nostore in ((1, 0), (0, 0), (1, 1), (0, 1)):
        ctx.set_option("bench_beta", beta); ctx.set_option("bulk_stagger", -1 if nostore else 0)
        r = {}
        for K in (256, 512):
            r[K] = round(float(np.median([ctx.bench_gemm_nt(7, 7936, 7936, K, True, 8208, 20)[1] for _ in range(3)])), 1)
        a = 2 * r[256] - r[512]; b = (r[512] - r[256]) / 256
        print(f"beta={beta} nostore={nostore} us(K=256)={r[256]} us(K=512)={r[512]} fixed a={a:.1f} us slope={b:.3f} us/K -> loop {7936*7937/b/1e6:.1f} TF/s", flush=True)
