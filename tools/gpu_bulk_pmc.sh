#!/bin/bash
# PMC passes of the bulk tile's K loop in isolation (bulk_probe = 3: C read and store removed; K = 2048, M = 8192, lower-triangular):
# LDS conflicts, wait cycles, MFMA-busy -- one --pmc group per run, --kernel-trace only.  Stops at the first failing step.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
stop() { echo "STOPPED after '$1' (rc=$2)"; exit "$2"; }
Q="rocprofv3 --kernel-trace --output-format csv"
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  for probe in 3 0; do
    K=$([ $probe = 3 ] && echo 2048 || echo 256)
    rm -rf gpurun_out/pmc_loop_${i}_p${probe}
    GSUM_BULK_PROBE=$probe timeout -k 10 120 $Q --pmc $grp -d gpurun_out/pmc_loop_${i}_p${probe} -- python3 tools/prof_gemm.py 7 8192 $K 1 3 > gpurun_out/pmc_loop_${i}_p${probe}.log 2>&1
    rc=$?; echo "group $i probe $probe rc=$rc"; [ $rc -eq 0 ] || stop "group $i probe $probe" $rc
  done
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_loop_*_p*")):
    if not d.split("/")[-1].startswith("pmc_loop_") or d.endswith(".log"):
        continue
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_gemm_ld3" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d.split("/")[-1], {k: v[-1] for k, v in agg.items()})
PY
