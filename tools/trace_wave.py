#!/usr/bin/env python3
"""Per-dispatch view of the LAST batch call in a rocprofv3 kernel trace (tools/prof_wave.py): kernel, start, duration, in time order.
    python tools/trace_wave.py <dir with *_kernel_trace.csv> [max rows]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call starts at the last k_build2 burst preceded by a gap
names = [r["Kernel_Name"] for r in rows]
last_begin = max(i for i, nme in enumerate(names) if nme.startswith("void k_build2") and (i == 0 or not names[i - 1].startswith("void k_build2")) and
                 sum(1 for x in names[i:] if x.startswith("k_finalize_g")) >= 1)
# take the last group of builds that still has finalize after it, then walk back to the first build of that call
sel = rows[last_begin:]
t0 = int(sel[0]["Start_Timestamp"])
tot = {}
for r in sel:
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot.setdefault(nm, [0, 0.0])
    tot[nm][0] += 1
    tot[nm][1] += dur
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 400
for r in sel[:lim]:
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:10.1f} us  +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f}  {nm:28s} grid {r.get("Grid_Size_X", r.get("Grid_Size", "?"))} q{r.get("Queue_Id", "?")}')
print("---- totals (count, sum us)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:28s} {v[0]:5d} {v[1]:10.1f}")
print("span us", (max(int(r["End_Timestamp"]) for r in sel) - t0) / 1e3)
