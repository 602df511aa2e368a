import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
import gsum_amd
from sklearn.gaussian_process.kernels import Matern, WhiteKernel
n = 16384
side = np.array([0.35, 0.65]) * np.sqrt(n)
X = np.random.RandomState(0).rand(n, 2) * side
y = np.random.RandomState(2).randn(n, 8)
kern = Matern(length_scale=[0.7, 1.3], nu=2.5) + WhiteKernel(1e-6, noise_level_bounds="fixed")
gp = gsum_amd.ConjugateGaussianProcess(kernel=kern, center=0, disp=0, df=1, scale=1, optimizer=None)
ctx = gsum_amd.default_context(0)
for i in range(5):
    t0 = time.perf_counter(); gp.fit(X, y); dt = time.perf_counter() - t0
    print(f"fit {i}: {dt*1e3:.1f} ms aborts={ctx.get_option('chain_aborts')} persist={ctx.get_option('chain_persist')} probe={ctx.get_option('chain_probe')}", flush=True)
