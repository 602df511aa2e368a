#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m gsum_amd.build
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "medium or grid" 2>&1 | tail -4 || exit 1
timeout -k 10 500 python -c "
import sys, time; sys.path.insert(0,'.')
import numpy as np, gsum_amd
from sklearn.gaussian_process.kernels import RBF
ctx = gsum_amd.default_context(0)
for n in (256, 512, 1024, 2048, 4096):
    X = 0.1*np.arange(n)[:,None]; Z = np.concatenate([np.random.RandomState(0).randn(n,6), np.ones((n,1))],1)
    ctx.set_inputs(X, Z)
    cnt = 1024 if n <= 2048 else 256
    descs = [gsum_amd.describe_kernel(RBF(0.2*(1+0.0005*i)), 1) for i in range(cnt)]
    ctx.set_option('medium_min_batch', 1); ctx.lml_resident(descs, 1e-10)
    t0=time.perf_counter(); a = ctx.lml_resident(descs, 1e-10); tm=(time.perf_counter()-t0)
    print(n, cnt, 'medium %.4f ms/eval (%.0f evals/s, %.1f TF/s)' % (tm/cnt*1e3, cnt/tm, cnt*n**3/3/tm/1e12), flush=True)
"
