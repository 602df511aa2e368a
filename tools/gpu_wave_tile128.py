"""Batches at large orders with the far updates of big trailing matrices on the 128 x 128 tile (option wave_tile128_rows) against the 128 x 64
tile throughout: evaluations per second and bit-identity.  -> profiles/r05_tile128.log"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsum_amd
from sklearn.gaussian_process.kernels import RBF
lab = gsum_amd.lab_context(0)
for n, N in ((12288, 20), (16384, 20), (8192, 20)):
    X = 0.1 * np.arange(n)[:, None]
    Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
    lab.set_inputs(X, Z)
    descs = lab.desc_array([gsum_amd.describe_kernel(RBF(float(e)), 1) for e in np.linspace(0.19, 0.21, N)])
    ref = None
    for rows in (0, 10240, 8192, 6144, 0, 10240):
        lab.set_option("wave_tile128_rows", rows)
        got = lab.lml_resident(descs, 1e-10)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); got = lab.lml_resident(descs, 1e-10); ts.append(time.perf_counter() - t0)
        same = ref is None or all(np.array_equal(a, b) for a, b in zip(got, ref))
        ref = ref or got
        print(f"n={n} {N} per call, 128x128 from {rows} rows: {N / min(ts):.2f} evals/s ({N * n ** 3 / 3 / min(ts) / 1e12:.1f} TF/s) identical={same}", flush=True)
    lab.set_option("release_scratch", 1)
lab.set_option("wave_tile128_rows", 0)
