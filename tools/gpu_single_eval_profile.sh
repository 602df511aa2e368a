#!/bin/bash
# refresh: probe (stage timings, batch sweep, diag stamps) + single-eval kernel trace stats + PCIe-inclusive rate
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m gsum_amd.build
timeout -k 10 400 python tools/gpu_probe.py > gpurun_out/probe.log 2>&1; echo rc=$?
grep "batch_\|^8192 1 1\|stamps" gpurun_out/probe.log | tail -14
rm -rf gpurun_out/prof_r1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1 -- python3 tools/prof_eval.py 2048 8192 > gpurun_out/rocprof.log 2>&1; echo rc=$?
timeout -k 10 200 python -c "
import sys, time; sys.path.insert(0,'.')
import numpy as np, gsum_amd
from sklearn.gaussian_process.kernels import RBF
ctx = gsum_amd.default_context(0)
desc = gsum_amd.describe_kernel(RBF(0.2), 1)
n, r = 8192, 6
X = 0.1*np.arange(n)[:,None]; Z = np.concatenate([np.random.RandomState(0).randn(n,r), np.ones((n,1))],1)
ctx.set_inputs(X, Z); ctx.lml_resident([desc], 1e-10)
for name, fn in (('resident', lambda: ctx.lml_resident([desc], 1e-10)), ('host inputs (PCIe-inclusive)', lambda: ctx.lml_batch([desc], X, Z, 1e-10))):
    fn(); t0=time.perf_counter()
    for _ in range(10): fn()
    print(name, 'ms/eval %.3f' % ((time.perf_counter()-t0)/10*1e3), flush=True)
"
