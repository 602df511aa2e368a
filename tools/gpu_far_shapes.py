import sys, os
sys.path.insert(0, '/root/repo')
import gsum_amd
lab = gsum_amd.lab_context(0)
for M, lda in ((7184, 8208), (7184, 16400), (11024, 16400), (15120, 16400)):
    for K in (256, 512, 1024):
        lab.bench_gemm_nt(7, M, M, K, True, lda, 3)
        tf, us = lab.bench_gemm_nt(7, M, M, K, True, lda, 12)
        print(f"tri M={M} lda={lda} K={K}: {tf:.1f} TF/s {us:.0f} us", flush=True)
