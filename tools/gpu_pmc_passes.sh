#!/bin/bash
# PMC passes for the shipped bulk kernel (k_gemm_ld3, exclusive launch) + bench under rocprof + bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m gsum_amd.build
rm -rf gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3 gpurun_out/pmc4 gpurun_out/pmc5 gpurun_out/prof_bench
P="python3 tools/prof_gemm.py 7 8192 256 1 3"
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc1 -- $P > gpurun_out/pmc1.log 2>&1; echo rc=$?
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc2 -- $P > gpurun_out/pmc2.log 2>&1; echo rc=$?
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc3 -- $P > gpurun_out/pmc3.log 2>&1; echo rc=$?
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc4 -- $P > gpurun_out/pmc4.log 2>&1; echo rc=$?
grep "^5 " gpurun_out/pmc1.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 18 --warmup 3 --cpu-evals 0 > gpurun_out/rocprof_bench.log 2>&1; echo rc=$?
timeout -k 10 600 python bench.py > gpurun_out/bench.log 2>&1; echo rc=$?
grep "^{" gpurun_out/bench.log | cut -c1-120
