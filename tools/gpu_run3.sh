#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m gsum_amd.build
timeout -k 10 300 python -c "
import sys; sys.path.insert(0,'.')
import gsum_amd, numpy as np
ctx = gsum_amd.default_context(0)
shapes = ((8192,8192,256,0,8208),(8192,8192,256,1,8208),(4096,4096,256,1,8208))
for M,N,K,tri,lda in shapes: ctx.bench_gemm_nt(0,M,N,K,bool(tri),lda,6)   # warm-up
res = {}
for rnd in range(4):
    for st in (0, 16):
        ctx.set_option('stagger', st)
        for cfg in (0, 5):
            for sh in shapes:
                M,N,K,tri,lda = sh
                res.setdefault((st,cfg,sh[:4]), []).append(ctx.bench_gemm_nt(cfg,M,N,K,bool(tri),lda,4)[0])
for k,v in sorted(res.items()): print(k, ['%.1f'%x for x in v], 'median %.1f' % np.median(v), flush=True)
" 2>&1 | tail -13
