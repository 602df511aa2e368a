#!/bin/bash
# round-3 profile collection: the bench line on the driver's command; rocprofv3 kernel-trace stats of the bench, of single
# evaluations (persistent-chain schedule and, for comparison, the host-enqueued one) and of the predict leg; the chain's own
# timeline; PMC passes (one --pmc group per run, --kernel-trace only) of the bulk GEMM; the schedule / gradient / bulk-tile checks.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
# a GPU step that fails, times out or is killed ends the script: no further GPU step is started behind it in the same call
stop() { echo "STOPPED after '$1' (rc=$2): later steps not run"; exit "$2"; }
python -m gsum_amd.build
PART="${1:-all}"          # 1: bench + kernel-trace profiles, 2: schedule checks + PMC passes, all: both (fits gpurun's 1200 s only when nothing is slow)
if [ "$PART" != 2 ]; then
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; rc=$?; echo "bench rc=$rc"; [ $rc -eq 0 ] || stop "bench" $rc
grep "^{" gpurun_out/bench.log | cut -c1-120
R="rocprofv3 --kernel-trace --stats --output-format csv"
rm -rf gpurun_out/prof_bench gpurun_out/prof_single gpurun_out/prof_single_host gpurun_out/prof_predict gpurun_out/pmc_*
timeout -k 10 400 $R -d gpurun_out/prof_bench -- python3 bench.py --steps 20 --warmup 3 --cpu-evals 0 --extras 0 --repeats 3 > gpurun_out/rocprof_bench.log 2>&1; rc=$?; echo "rocprof bench rc=$rc"; [ $rc -eq 0 ] || stop "rocprof bench" $rc
timeout -k 10 200 $R -d gpurun_out/prof_single -- python3 tools/prof_eval.py 2048 8192 > gpurun_out/rocprof_single.log 2>&1; rc=$?; echo "single rc=$rc"; [ $rc -eq 0 ] || stop "single" $rc
GSUM_PROF_CHAIN=0 timeout -k 10 200 $R -d gpurun_out/prof_single_host -- python3 tools/prof_eval.py 8192 > gpurun_out/rocprof_single_host.log 2>&1; rc=$?; echo "single (host-enqueued) rc=$rc"; [ $rc -eq 0 ] || stop "single (host-enqueued)" $rc
timeout -k 10 200 $R -d gpurun_out/prof_predict -- python3 bench.py --config predict > gpurun_out/rocprof_predict.log 2>&1; rc=$?; echo "predict rc=$rc"; [ $rc -eq 0 ] || stop "predict" $rc
fi
if [ "$PART" != 1 ]; then
timeout -k 10 200 python tools/gpu_chain_timeline.py 8192 512 > gpurun_out/chain_timeline.log 2>&1; rc=$?; echo "timeline rc=$rc"; [ $rc -eq 0 ] || stop "timeline" $rc
timeout -k 10 300 python tools/gpu_chain_check.py 5 > gpurun_out/chain_check.log 2>&1; rc=$?; echo "chain check rc=$rc"; [ $rc -eq 0 ] || stop "chain check" $rc
timeout -k 10 300 python tools/gpu_grad_batch.py > gpurun_out/grad_batch.log 2>&1; rc=$?; echo "grad batch rc=$rc"; [ $rc -eq 0 ] || stop "grad batch" $rc
timeout -k 10 300 python tools/gpu_bulk_stages.py > gpurun_out/bulk_stages.log 2>&1; rc=$?; echo "bulk stages rc=$rc"; [ $rc -eq 0 ] || stop "bulk stages" $rc
P="python3 tools/prof_gemm.py 7 8192 256 1 3"
Q="rocprofv3 --kernel-trace --output-format csv"
timeout -k 10 200 $Q --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES -d gpurun_out/pmc_gemm1 -- $P > gpurun_out/pmc_gemm1.log 2>&1; rc=$?; echo "step rc=$rc"; [ $rc -eq 0 ] || stop "step" $rc
timeout -k 10 200 $Q --pmc FETCH_SIZE -d gpurun_out/pmc_gemm2 -- $P > gpurun_out/pmc_gemm2.log 2>&1; rc=$?; echo "step rc=$rc"; [ $rc -eq 0 ] || stop "step" $rc
timeout -k 10 200 $Q --pmc WRITE_SIZE -d gpurun_out/pmc_gemm3 -- $P > gpurun_out/pmc_gemm3.log 2>&1; rc=$?; echo "step rc=$rc"; [ $rc -eq 0 ] || stop "step" $rc
fi
find gpurun_out -name "*kernel_stats.csv" | head; du -sh gpurun_out
