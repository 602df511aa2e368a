#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m gsum_amd.build
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -12
timeout -k 10 200 python -c "
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, gsum_amd
from conftest import load_golden
from sklearn.gaussian_process.kernels import RBF, WhiteKernel
g = load_golden('notebook_grid.json')
X, y = np.array(g['X_train']), np.array(g['y_train'])
kern = RBF(0.2) + WhiteKernel(g['nugget'], noise_level_bounds='fixed')
gp = gsum_amd.TruncationGP(kernel=kern, ref=g['ref'], ratio=0.5, center=0, disp=0, df=1, scale=1, optimizer=None)
gp.fit(X, y, orders=np.array(g['orders']))
thetas = [[t] for t in np.log(g['ls_vals'])]
for rep in range(2):
    t0 = time.perf_counter(); grid = gp.log_marginal_likelihood_grid(thetas, g['ratio_vals'], mode='full'); dt = time.perf_counter() - t0
    print('notebook 80x100 grid (n_train=5), full mode: %.1f ms, argmax' % (dt*1e3), np.unravel_index(np.argmax(grid), grid.shape), flush=True)
" 2>&1 | tail -3
