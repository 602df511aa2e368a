#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
which gdb rocgdb 2>&1 | head -2
timeout -k 10 120 python -X faulthandler -c "
import sys; sys.path.insert(0,'.')
import gsum_amd
ctx = gsum_amd.default_context(0)
print('ctx ok', flush=True)
print(ctx.probe_hbm_write(1<<28), flush=True)
for w,a in ((1,8),(1,1)):
    print('probe', w, a, flush=True)
    print(ctx.probe_mfma_f64(100, w, a), flush=True)
" > gpurun_out/dbg.log 2>&1; echo rc=$?; tail -30 gpurun_out/dbg.log
