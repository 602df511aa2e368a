"""Lab experiment (round 5): every far update on its group's chain stream (wave_far_own) instead of on the one shared bulk stream: the groups'
far launches overlap one another's ramp and drain; the chain kernels lose their priority over them."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsum_amd
from sklearn.gaussian_process.kernels import RBF
lab = gsum_amd.lab_context(0)
n = 8192
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
lab.set_inputs(X, Z)
for N in (20, 24, 96):
    descs = lab.desc_array([gsum_amd.describe_kernel(RBF(float(e)), 1) for e in np.linspace(0.19, 0.21, N)])
    ref = None
    for own, groups, size in ((0, 3, 8), (1, 3, 8), (1, 2, 10), (1, 4, 6), (0, 3, 8), (1, 3, 8)):
        lab.set_option("wave_far_own", own); lab.set_option("wave_groups", groups); lab.set_option("wave_size", size)
        got = lab.lml_resident(descs, 1e-10)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); got = lab.lml_resident(descs, 1e-10); ts.append(time.perf_counter() - t0)
        same = ref is None or all(np.array_equal(a, b) for a, b in zip(got, ref))
        ref = ref or got
        print(f"{N} per call, far_own={own} groups={groups}x{size}: {N / min(ts):.1f} evals/s (median {N / np.median(ts):.1f}) identical={same}", flush=True)
lab.set_option("wave_far_own", 0); lab.set_option("wave_groups", 3); lab.set_option("wave_size", 8)
