#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m gsum_amd.build
echo "=== smoke"; timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "=== 2 ranks on one GPU (gloo rehearsal)"
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 8 --warmup 2 --backend gloo --device 0 --cpu-evals 0 2>&1 | grep "^{" | cut -c1-400
echo "=== predict at scale"
timeout -k 10 500 python -c "
import sys, time; sys.path.insert(0,'.')
import numpy as np, gsum_amd
from sklearn.gaussian_process.kernels import Matern, WhiteKernel
from oracle import gsum_oracle as orc
for n, m in ((4096, 2048), (16384, 2048)):
    rng = np.random.RandomState(0)
    box = np.array([0.7, 1.3]) * np.sqrt(n) * 0.5
    X = rng.rand(n, 2) * box; Xs = rng.rand(m, 2) * box
    kern = Matern(length_scale=[0.7, 1.3], nu=2.5) + WhiteKernel(1e-6, noise_level_bounds='fixed')
    y = rng.randn(n, 8)
    gp = gsum_amd.ConjugateGaussianProcess(kernel=kern, optimizer=None)
    t0 = time.perf_counter(); gp.fit(X, y); t1 = time.perf_counter()
    mean, std = gp.predict(Xs, return_std=True); t2 = time.perf_counter()
    mean, std = gp.predict(Xs, return_std=True); t3 = time.perf_counter()
    print('n', n, 'm', m, 'fit %.3f s' % (t1-t0), 'predict %.3f s (2nd %.3f s)' % (t2-t1, t3-t2), 'std range', std.min(), std.max(), flush=True)
    if n <= 4096:
        fit = orc.cgp_fit(kern, X, y); mo, so = orc.cgp_predict(fit, Xs, return_std=True)
        print('  vs oracle: mean max abs', np.abs(mean-mo).max(), 'var max abs/cov_factor', np.abs(std**2-so**2).max()/fit['cov_factor'], flush=True)
" 2>&1 | tail -6
