import os, sys
sys.path.insert(0, '/root/repo')
import gsum_amd
lab = gsum_amd.lab_context(0)
lab.bench_gemm_nt(7, 7184, 7184, 1024, True, 8208, 5)
for tri, M, N, K, lda in ((0, 7184, 7184, 1024, 8208), (1, 7184, 7184, 1024, 8208), (0, 7168, 7168, 1024, 8208), (1, 7168, 7168, 1024, 8208), (1, 8192, 8192, 1024, 8208),
                          (0, 8192, 8192, 1024, 8208), (0, 7184, 3592, 1024, 8208), (0, 3592, 7184, 1024, 8208), (0, 7184, 7184, 1024, 8208), (1, 7184, 7184, 1024, 8208)):
    lab.bench_gemm_nt(7, M, N, K, bool(tri), lda, 3)
    tf, us = lab.bench_gemm_nt(7, M, N, K, bool(tri), lda, 16)
    print(f"tri={tri} M={M} N={N} K={K}: {tf:.1f} TF/s {us:.0f} us", flush=True)
