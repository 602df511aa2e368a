import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
import gsum_amd
from sklearn.gaussian_process.kernels import RBF
ctx = gsum_amd.lab_context(0)
for n in (1024, 2048, 4096):
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, 4), np.ones((n, 1))], axis=1)
    ctx.set_inputs(X, Z)
    ctx.set_option("medium_min_batch", 1)
    for N in (128, 256, 512, 1024):
        descs = ctx.desc_array([gsum_amd.describe_kernel(RBF(float(e)), 1) for e in np.linspace(0.15, 0.25, N)])
        ctx.lml_resident(descs, 1e-10)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); ctx.lml_resident(descs, 1e-10); ts.append(time.perf_counter() - t0)
        print(f"n={n} {N} evals: {min(ts)*1e3:.2f} ms  {N/min(ts):.0f} evals/s  {N*n**3/3/min(ts)/1e12:.1f} TF/s", flush=True)
    ctx.set_option("release_scratch", 1)
