#!/usr/bin/env python3
"""Copy what the GPU-box scripts left under gpurun_out/ (scratch) into profiles/ (tracked) as the round's evidence:
kernel-trace statistics, bench lines, PMC passes condensed to JSON, RCCL logs, parity-vs-truth numbers.
    python tools/collect_profiles.py [round tag, default r02]"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT, SRC = os.path.join(ROOT, "profiles"), os.path.join(ROOT, "gpurun_out")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"


def newest(sub, pattern):
    files = sorted(glob.glob(os.path.join(SRC, sub, "**", pattern), recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


def _reference_time():
    """When the run being collected happened: the newest bench log under gpurun_out/."""
    logs = [os.path.join(SRC, f) for f in ("bench.log", "rocprof_bench.log") if os.path.exists(os.path.join(SRC, f))]
    return max(map(os.path.getmtime, logs)) if logs else None


RUN_TIME = _reference_time()
MAX_AGE_S = 2 * 3600.0


def stale(src):
    """gpurun_out/ is scratch that accumulates over a session: a file much older than the run being collected belongs to an earlier
    run (or round) and must not be relabelled with this round's tag."""
    return RUN_TIME is not None and os.path.getmtime(src) < RUN_TIME - MAX_AGE_S


def copy(src, name):
    if src and os.path.exists(src) and stale(src):
        print("SKIPPED (older than this run):", os.path.relpath(src, ROOT), "->", f"{TAG}_{name}")
        return
    if src and os.path.exists(src):
        shutil.copyfile(src, os.path.join(OUT, f"{TAG}_{name}"))
        print("wrote", f"{TAG}_{name}")


def json_line(log, name):
    path = os.path.join(SRC, log)
    if not os.path.exists(path):
        return
    if stale(path):
        print("SKIPPED (older than this run):", os.path.relpath(path, ROOT), "->", f"{TAG}_{name}")
        return
    lines = [ln for ln in open(path) if ln.startswith("{")]
    if lines:
        open(os.path.join(OUT, f"{TAG}_{name}"), "w").write(lines[-1])
        print("wrote", f"{TAG}_{name}")


def counters(sub, prefix):
    f = newest(sub, "*counter_collection.csv")
    agg = {}
    if not f:
        return agg
    if stale(f):
        print("SKIPPED (older than this run):", os.path.relpath(f, ROOT))
        return agg
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith(prefix):
            agg.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return agg


def last(agg, name):
    return agg[name][-1][0] if name in agg else None


for sub, name in (("prof_bench", "bench_kernel_stats.csv"), ("prof_single", "single_eval_n2048_n8192_kernel_stats.csv"),
                  ("prof_single_host", "single_eval_n8192_host_enqueued_kernel_stats.csv"),
                  ("prof_predict", "predict_kernel_stats.csv"), ("prof_grad", "grad_n8192_kernel_stats.csv")):
    copy(newest(sub, "*kernel_stats.csv"), name)
# the single-evaluation kernel trace itself (start / end of every dispatch) and the per-step timeline derived from it
tr = newest("prof_single", "*kernel_trace.csv")
if tr and not os.path.exists(os.path.join(SRC, "single_eval_chain_timeline.txt")):
    copy(tr, "single_eval_kernel_trace.csv")
    import subprocess
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_timeline.py"), tr, "60"], capture_output=True, text=True).stdout
    open(os.path.join(OUT, f"{TAG}_single_eval_timeline.txt"), "w").write(
        "Last n = 8192 factorisation of `rocprofv3 --kernel-trace -- python3 tools/prof_eval.py 2048 8192` (look-ahead schedule, one\n"
        "evaluation in flight); columns: start us, end us, duration us, kernel, stream, grid.  Then the start-to-start period of the\n"
        "bulk launches (= duration of an outer step of 256 columns) and the bulk launches' own durations.\n\n" + txt)
    print("wrote", f"{TAG}_single_eval_timeline.txt")
json_line("bench.log", "bench_default.jsonl")
json_line("rocprof_bench.log", "bench_under_rocprof.jsonl")
json_line("plain_n1.log", "plain_n1.jsonl")
json_line("torchrun_world1_nccl.log", "torchrun_world1_nccl.jsonl")
json_line("two_ranks_one_gpu_gloo.log", "two_ranks_one_gpu_gloo.jsonl")
for name in ("wave_trace_1x20.txt", "wave_trace_3x7.txt", "parity_achieved.json"):
    copy(os.path.join(SRC, name), name)
json_line("bench_predict.log", "bench_predict.jsonl")
copy(os.path.join(SRC, "single_eval_chain_timeline.txt"), "single_eval_timeline.txt")       # round 3: the chain kernel's own stamps
for log in ("rocprof_medium.log", "wave_profile.log", "wave_sweep.log", "chain_abort_repro.log", "chain_check.log", "grad_batch.log", "bulk_cphase.log", "bulk_stages.log", "rocprof_single.log", "rccl_world1.log", "two_ranks_one_gpu_gloo.log", "two_ranks_one_gpu_nccl.log", "medium_rates.log", "r2_kernels.log",
            "single_sweep.log", "slots_sweep.log", "clock_power.log", "medium_phases.log"):
    path = os.path.join(SRC, log)
    if os.path.exists(path):
        text = open(path, errors="replace").read().splitlines()
        keep = text if len(text) <= 260 else text[:200] + ["... (%d lines cut) ..." % (len(text) - 260)] + text[-60:]
        open(os.path.join(OUT, f"{TAG}_{log}"), "w").write("\n".join(keep) + "\n")
        print("wrote", f"{TAG}_{log}")
copy(os.path.join(SRC, "truth_errors.json"), "truth_errors.json")
copy(os.path.join(SRC, "r2_kernels.json"), "kernels_ab.json")

# ---- PMC: the dominant kernel's launches inside a batch call (k_gemm_ld3g; counter passes serialise dispatches, so each launch's
# counters are its own), and the one-product form of the same tile (exclusive launch M = 8192, K = 256) for continuity with rounds 1-3
def all_values(sub, prefix, counter):
    f = newest(sub, "*counter_collection.csv")
    if not f or stale(f):
        return []
    return [(float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(prefix) and r["Counter_Name"] == counter]


g = {}
for sub in ("pmc_gemm1", "pmc_gemm2", "pmc_gemm3", "pmc_gemm4"):
    g.update(counters(sub, "void k_gemm_ld3<"))
    g.update(counters(sub, "k_gemm_ld3<"))
one = None
if g:
    fetch, write = last(g, "FETCH_SIZE"), last(g, "WRITE_SIZE")
    dur_ns = g["GRBM_GUI_ACTIVE"][-1][1] if "GRBM_GUI_ACTIVE" in g else None
    busy, gui = last(g, "SQ_VALU_MFMA_BUSY_CYCLES"), last(g, "GRBM_GUI_ACTIVE")
    one = {"kernel": "k_gemm_ld3<2>", "command": "rocprofv3 --pmc <group> --kernel-trace -- python3 tools/prof_gemm.py 7 8192 256 1 3 (one pass per counter group)",
           "shape": {"M": 8192, "N": 8192, "K": 256, "tri": 1, "tiles": 4160},
           "counters": {k: v[-1][0] for k, v in g.items()}, "duration_us_profiled": dur_ns / 1e3 if dur_ns else None,
           "hbm_traffic_bytes_per_launch": (fetch * 2048 + write * 1024) if fetch and write else None, "algorithmic_bytes_per_launch": 561000000,
           "mfma_busy_frac_profiled": busy / (gui / 8 * 1024) if busy and gui else None}
sys.path.insert(0, os.path.join(ROOT, "tools"))
from wave_plan import far_launches  # noqa: E402
plan = far_launches(8192, [7, 7, 6])
fetch = all_values("pmc_wave3", "k_gemm_ld3g", "FETCH_SIZE")
write = all_values("pmc_wave4", "k_gemm_ld3g", "WRITE_SIZE")
busy = all_values("pmc_wave1", "k_gemm_ld3g", "SQ_VALU_MFMA_BUSY_CYCLES")
gui = all_values("pmc_wave1", "k_gemm_ld3g", "GRBM_GUI_ACTIVE")
if fetch and write:
    calls = len(fetch) / len(plan)
    rd, wr = sum(v for v, _ in fetch) * 2048 / len(fetch), sum(v for v, _ in write) * 1024 / len(write)
    alg = sum(x["bytes"] for x in plan) / len(plan)
    rec = {"kernel": "k_gemm_ld3g",
           "what": "HBM bytes per launch of the batch's trailing-update launches, averaged over the %d k_gemm_ld3g dispatches of %g calls of "
                   "20 evaluations (3 groups of 7, 7, 6; n = 8192): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes "
                   "(tools/gpu_r4_profiles.sh; FETCH_SIZE x 2 KiB -- gfx950 counts 64 B per 128-B request --, WRITE_SIZE x 1 KiB)" % (len(fetch), calls),
           "command": "rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 tools/prof_wave.py 3 8 4",
           "launches_per_call": len(plan), "dispatches_counted": len(fetch),
           "derived": {"hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_traffic_bytes_per_launch": rd + wr,
                       "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (rd + wr) / alg,
                       "algorithmic_flops_per_launch": sum(x["flops"] for x in plan) / len(plan),
                       "mfma_busy_frac_profiled": (sum(v for v, _ in busy) / (sum(v for v, _ in gui) / 8 * 1024)) if busy and gui else None,
                       "avg_launch_us_profiled": sum(d for _, d in fetch) / len(fetch) / 1e3,
                       "note": "algorithmic bytes of a launch = for every member its C lower triangle read and written once (16 B x M (M + 1) / 2) + its "
                               "M x K panel rows read once (tools/wave_plan.py restates the launch list); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES "
                               "(summed over all SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), both summed over the launches (serialised by the "
                               "counter pass: each launch alone on the chip)"},
           "one_product_K256": one}
    json.dump(rec, open(os.path.join(OUT, f"{TAG}_gemm_pmc.json"), "w"), indent=1)
    print("wrote", f"{TAG}_gemm_pmc.json")
elif one:
    json.dump({"kernel": "k_gemm_ld3<2>", "derived": {"hbm_traffic_bytes_per_launch": one["hbm_traffic_bytes_per_launch"],
                                                      "algorithmic_bytes_per_launch": 561000000}, "one_product_K256": one},
              open(os.path.join(OUT, f"{TAG}_gemm_pmc.json"), "w"), indent=1)
    print("wrote", f"{TAG}_gemm_pmc.json (one-product launch only)")
b = {}
for sub in ("pmc_build1", "pmc_build2", "pmc_build3"):
    b.update(counters(sub, "void k_build2"))
if b:
    fetch, write = last(b, "FETCH_SIZE"), last(b, "WRITE_SIZE")
    dur = [d for _, d in b.get("WRITE_SIZE", b.get("GRBM_GUI_ACTIVE", []))]
    rec = {"kernel": "k_build2<false, RBF, 1-D>", "command": "rocprofv3 --pmc <group> --kernel-trace -- python3 tools/prof_build.py 8192 3",
           "counters": {k: [x for x, _ in v] for k, v in b.items()}, "durations_us_profiled": [d / 1e3 for d in dur],
           "derived": {"hbm_write_bytes": write * 1024 if write else None, "hbm_read_bytes": fetch * 2048 if fetch else None,
                       "algorithmic_bytes": 4 * 8192 * 8192 + 4 * 8192 * 128,
                       "valu_instructions_per_entry": (last(b, "SQ_INSTS_VALU") * 64 / (2080 * 128 * 128)) if "SQ_INSTS_VALU" in b else None,
                       "note": "WRITE_SIZE x 1 KiB equals the bytes the kernel stores (lower 128-column tiles); nothing is read back"}}
    json.dump(rec, open(os.path.join(OUT, f"{TAG}_build_pmc.json"), "w"), indent=1)
    print("wrote", f"{TAG}_build_pmc.json")

# ---- round 5: k_lml_medium (BASELINE configs[1]: n = 2048, 512 evaluations per launch, one workgroup each) under PMC
def any_values(sub, needle, counter):
    f = newest(sub, "*counter_collection.csv")
    if not f or stale(f):
        return []
    return [(float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            for r in csv.DictReader(open(f)) if needle in r["Kernel_Name"] and r["Counter_Name"] == counter]


med = {c: any_values(sub, "k_lml_medium", c) for sub, cs in (("pmc_med1", ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVES")),
                                                                ("pmc_med2", ("FETCH_SIZE",)), ("pmc_med3", ("WRITE_SIZE",)),
                                                                ("pmc_med4", ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                                                              "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU_MFMA_MOPS_F64"))) for c in cs}
if med.get("FETCH_SIZE") and med.get("WRITE_SIZE"):
    n_, evals = 2048, 512
    last_of = lambda c: med[c][-1] if med.get(c) else (None, None)      # noqa: E731  (the last dispatch: warmed-up buffers)
    fetch, dur_f = last_of("FETCH_SIZE")
    write, _ = last_of("WRITE_SIZE")
    gui, dur_g = last_of("GRBM_GUI_ACTIVE")
    busy, _ = last_of("SQ_VALU_MFMA_BUSY_CYCLES")
    rd, wr = fetch * 2048, write * 1024
    flops = evals * n_ ** 3 / 3.0
    dur = (dur_g or dur_f) / 1e9
    rec = {"kernel": "k_lml_medium<false>", "command": "rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 tools/prof_medium.py 2048 512 2",
           "what": "one launch = 512 full evaluations at n = 2048 (K build + Cholesky + solve + Gram), one workgroup each, two per CU; counters of the "
                   "last dispatch of each pass (FETCH_SIZE x 2 KiB, WRITE_SIZE x 1 KiB as the guide prescribes for gfx950)",
           "counters": {c: (v[-1][0] if v else None) for c, v in med.items()},
           "derived": {"duration_ms_profiled": dur * 1e3, "evals_per_s_profiled": evals / dur, "cholesky_tflops": flops / dur / 1e12,
                       "frac_of_fp64_mfma_peak": flops / dur / 1e12 / 78.6,
                       "hbm_read_bytes_per_eval": rd / evals, "hbm_write_bytes_per_eval": wr / evals,
                       "fabric_side_GBps": (rd + wr) / dur / 1e9, "frac_of_hbm_peak_8TBps": (rd + wr) / dur / 8e12,
                       "matrix_bytes_per_eval": n_ * (n_ + 16) * 8,
                       "flops_per_fabric_byte": flops / (rd + wr),
                       "mfma_busy_frac": (busy / (gui / 8 * 1024)) if busy and gui else None,
                       "note": "memory-busy = fabric-side bytes / time against the 8 TB/s spec (an in-order sweep measures 6.0-6.3 TB/s on this part); "
                               "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)"}}
    json.dump(rec, open(os.path.join(OUT, f"{TAG}_medium_pmc.json"), "w"), indent=1)
    print("wrote", f"{TAG}_medium_pmc.json")
copy(newest("prof_medium", "*kernel_stats.csv"), "medium_n2048_kernel_stats.csv")
json_line("inproc1.log", "inproc1.jsonl")
json_line("inproc2_one_gpu.log", "inproc2_two_contexts_one_gpu.jsonl")
