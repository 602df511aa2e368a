#!/usr/bin/env python3
"""Timeline of the last n = 8192 factorisation in a rocprofv3 --kernel-trace CSV:  trace_timeline.py <kernel_trace.csv> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 45
idx = [i for i, r in enumerate(rows) if "k_build2" in r["Kernel_Name"]]
ev = rows[idx[-1]:]
t0 = int(ev[0]["Start_Timestamp"])
NAMES = ["k_potrf_diag256", "k_potrf_diag", "k_gemm_ld3", "k_gemm_nt", "k_panel256", "k_panel", "k_build2", "k_finalize", "k_save_diag",
         "k_set_border", "copyBuffer", "fillBuffer"]


def short(n):
    for k in NAMES:
        if k in n:
            return k
    return n[:20]


out = [((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, short(r["Kernel_Name"]), r["Stream_Id"],
        r["Grid_Size_X"]) for r in ev]
for o in out[:nshow]:
    print("%8.1f %8.1f %6.1f  %-16s s%s grid=%s" % (o[0], o[1], o[1] - o[0], o[2], o[3], o[4]))
print("...")
for o in out[-16:]:
    print("%8.1f %8.1f %6.1f  %-16s s%s grid=%s" % (o[0], o[1], o[1] - o[0], o[2], o[3], o[4]))
# per outer step: start of consecutive bulk launches
bulk = [o for o in out if o[2] == "k_gemm_ld3"]
print("bulk launch starts (us):", " ".join("%.0f" % b[0] for b in bulk))
print("periods:", " ".join("%.0f" % (b[0] - a[0]) for a, b in zip(bulk, bulk[1:])))
print("bulk durations:", " ".join("%.0f" % (b[1] - b[0]) for b in bulk))
