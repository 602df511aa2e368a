#!/bin/bash
# round-2 profile collection: rocprofv3 kernel-trace stats of the bench, of single evaluations, of the predict leg and of
# value + gradient; PMC passes (one --pmc group per run, --kernel-trace only) of the bulk GEMM and of the kernel build.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m gsum_amd.build
R="rocprofv3 --kernel-trace --stats --output-format csv"
rm -rf gpurun_out/prof_bench gpurun_out/prof_single gpurun_out/prof_predict gpurun_out/prof_grad gpurun_out/pmc_*
timeout -k 10 400 $R -d gpurun_out/prof_bench -- python3 bench.py --steps 20 --warmup 3 --cpu-evals 0 --extras 0 --repeats 3 > gpurun_out/rocprof_bench.log 2>&1; echo "bench rc=$?"
grep "^{" gpurun_out/rocprof_bench.log | cut -c1-160
timeout -k 10 200 $R -d gpurun_out/prof_single -- python3 tools/prof_eval.py 2048 8192 > gpurun_out/rocprof_single.log 2>&1; echo "single rc=$?"
timeout -k 10 200 $R -d gpurun_out/prof_predict -- python3 bench.py --config predict > gpurun_out/rocprof_predict.log 2>&1; echo "predict rc=$?"
timeout -k 10 200 $R -d gpurun_out/prof_grad -- python3 tools/prof_grad.py 8192 3 > gpurun_out/rocprof_grad.log 2>&1; echo "grad rc=$?"
P="python3 tools/prof_gemm.py 7 8192 256 1 3"
Q="rocprofv3 --kernel-trace --output-format csv"
timeout -k 10 200 $Q --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES -d gpurun_out/pmc_gemm1 -- $P > gpurun_out/pmc_gemm1.log 2>&1; echo rc=$?
timeout -k 10 200 $Q --pmc FETCH_SIZE -d gpurun_out/pmc_gemm2 -- $P > gpurun_out/pmc_gemm2.log 2>&1; echo rc=$?
timeout -k 10 200 $Q --pmc WRITE_SIZE -d gpurun_out/pmc_gemm3 -- $P > gpurun_out/pmc_gemm3.log 2>&1; echo rc=$?
timeout -k 10 200 $Q --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d gpurun_out/pmc_gemm4 -- $P > gpurun_out/pmc_gemm4.log 2>&1; echo rc=$?
B="python3 tools/prof_build.py 8192 3"
timeout -k 10 200 $Q --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VALU -d gpurun_out/pmc_build1 -- $B > gpurun_out/pmc_build1.log 2>&1; echo rc=$?
timeout -k 10 200 $Q --pmc WRITE_SIZE -d gpurun_out/pmc_build2 -- $B > gpurun_out/pmc_build2.log 2>&1; echo rc=$?
timeout -k 10 200 $Q --pmc FETCH_SIZE -d gpurun_out/pmc_build3 -- $B > gpurun_out/pmc_build3.log 2>&1; echo rc=$?
find gpurun_out -name "*kernel_stats.csv" | head; du -sh gpurun_out
