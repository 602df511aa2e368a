"""Leading-dimension sweep of the bulk tile alone (lab microbenchmark): lower-triangle update at M = 7184 (the n = 8192 trailing
matrix at outer step 3) for K = 256 / 1024 and row strides around and beyond n + 16.  -> profiles/r05_lda_sweep.log"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsum_amd
lab = gsum_amd.lab_context(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 7184
for lda in (8208, 8192 + 32, 8192 + 48, 8192 + 64, 8192 + 80, 8192 + 144, 8192 + 272, 8192 + 528, 8192 + 1040, 9232 + 1024, 12304, 16400, 16384 + 32, 16384 + 528):
    row = []
    for K in (256, 1024):
        lab.bench_gemm_nt(7, M, M, K, True, lda, 3)
        tf, us = lab.bench_gemm_nt(7, M, M, K, True, lda, 16)
        row.append(f"K={K}: {tf:5.1f} TF/s")
    print(f"M={M} lda={lda:6d} ({lda * 8 / 1024:8.3f} KiB)  " + "   ".join(row), flush=True)
