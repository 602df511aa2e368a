"""The 128 x 128 bulk tile (cfg 8, round 5) against the 128 x 64 one (cfg 7) alone on the chip: lower-triangle updates (one matrix) and
rectangles at the shapes the schedules launch.  -> profiles/r05_tile128.log"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsum_amd
lab = gsum_amd.lab_context(0)
lab.bench_gemm_nt(7, 7184, 7184, 1024, True, 8208, 5)
for tri, M, N, K, lda in ((1, 7184, 7184, 1024, 8208), (1, 7184, 7184, 256, 8208), (1, 4112, 4112, 1024, 8208), (1, 11280, 11280, 1024, 12304), (1, 15120, 15120, 1024, 16400),
                          (1, 15120, 15120, 512, 16400), (0, 8192, 8192, 1024, 8208), (0, 2048, 14000, 512, 16400)):
    r = []
    for cfg in (7, 8, 7, 8):
        lab.bench_gemm_nt(cfg, M, N, K, bool(tri), lda, 3)
        r.append(lab.bench_gemm_nt(cfg, M, N, K, bool(tri), lda, 12)[0])
    print(f"tri={tri} M={M} N={N} K={K}: 128x64 {r[0]:.1f} / {r[2]:.1f} TF/s   128x128 {r[1]:.1f} / {r[3]:.1f} TF/s", flush=True)
