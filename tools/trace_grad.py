#!/usr/bin/env python3
"""Timeline of the LAST value + gradient evaluation in a rocprofv3 --kernel-trace CSV (tools/prof_grad.py):  trace_grad.py <dir or csv>
Prints, per kernel class, first start / last end / summed duration relative to the evaluation's kernel build, and the tail of the launch list."""
import csv
import glob
import os
import sys

p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_build" in r["Kernel_Name"]]
ev = rows[idx[-1]:]
t0 = int(ev[0]["Start_Timestamp"])


def short(n):
    n = n.split("(")[0]
    return n.replace("void ", "")[:34]


cls = {}
for r in ev:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    c = cls.setdefault(short(r["Kernel_Name"]), [s, e, 0.0, 0])
    c[0], c[1], c[2], c[3] = min(c[0], s), max(c[1], e), c[2] + e - s, c[3] + 1
print("%-36s %9s %9s %9s %6s" % ("kernel", "first us", "last us", "sum us", "calls"))
for k, c in sorted(cls.items(), key=lambda kv: kv[1][0]):
    print("%-36s %9.1f %9.1f %9.1f %6d" % (k, c[0], c[1], c[2], c[3]))
print("... last launches")
for r in ev[-14:]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%9.1f %9.1f %8.1f  %-34s s%s grid=%s" % (s, e, e - s, short(r["Kernel_Name"]), r.get("Stream_Id", "?"), r["Grid_Size_X"]))
