#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m gsum_amd.build
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gemm" 2>&1 | tail -2
timeout -k 10 200 python -c "
import sys; sys.path.insert(0,'.')
import gsum_amd
ctx = gsum_amd.default_context(0)
for M,K in ((8192,256),(1024,256)):
    print(M,K,{k: round(v) for k,v in ctx.debug_gemm_phases(M,K,8208).items()}, flush=True)
for cfg in (0,3,1):
    for (M,N,K,tri,lda) in ((8192,8192,256,1,8208),(8192,8192,256,0,8208),(4096,4096,256,1,8208),(8192,256,256,0,8208),(8192,128,128,0,8208)):
        if cfg==1 and tri: continue
        print('cfg',cfg,'M',M,'N',N,'K',K,'tri',tri,'lda',lda, 'TF/s %.1f  us %.1f' % ctx.bench_gemm_nt(cfg,M,N,K,bool(tri),lda,4), flush=True)
" 2>&1 | tail -16
