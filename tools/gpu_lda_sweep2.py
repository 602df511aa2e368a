"""Row stride of the workspace matrices: the bulk tile alone (lab microbenchmark, lower-triangle update at M = np - 1024, K = 256 and
1024) for ld = np + pad over the padded orders the schedules use.  -> profiles/r05_lda_sweep.log"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsum_amd
lab = gsum_amd.lab_context(0)
pads = (16, 32, 48, 64, 80, 112, 144, 272)
print("TF/s at K = 256 | K = 1024 for ld = np + pad;  pads:", pads, flush=True)
for np_ in (2048, 3072, 4096, 5120, 6144, 7168, 8192, 10240, 12288, 16384):
    M = np_ - 1024 + 16
    cells = []
    for pad in pads:
        lda = np_ + pad
        v = []
        for K in (256, 1024):
            lab.bench_gemm_nt(7, M, M, K, True, lda, 3)
            tf, us = lab.bench_gemm_nt(7, M, M, K, True, lda, 24 if np_ <= 6144 else 10)
            v.append(tf)
        cells.append(f"{v[0]:4.1f}|{v[1]:4.1f}")
    print(f"np={np_:6d} M={M:6d}  " + "  ".join(cells), flush=True)
