#!/bin/bash
# round-5 evidence: the bench line on the driver's command, the same under rocprofv3 --kernel-trace --stats, under torch.distributed.run
# (world 1, nccl), through the in-library device group (--inproc 1; --inproc 2 on two contexts of this one GPU); PMC passes (one counter
# group per run, --kernel-trace only) of the batch's launches, of the one-product bulk tile and of k_lml_medium; single-evaluation and
# predict kernel statistics.    tools/gpu_r5_profiles.sh [1|2|all]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
stop() { echo "STOPPED after '$1' (rc=$2): later steps not run"; exit "$2"; }
python -m gsum_amd.build --lab > /dev/null
PART="${1:-all}"
R="rocprofv3 --kernel-trace --stats --output-format csv"
Q="rocprofv3 --kernel-trace --output-format csv"
if [ "$PART" != 2 ]; then
  timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; rc=$?; echo "bench rc=$rc"; [ $rc -eq 0 ] || stop "bench" $rc
  grep "^{" gpurun_out/bench.log | cut -c1-140
  rm -rf gpurun_out/prof_bench gpurun_out/tw3
  timeout -k 10 300 $R -d gpurun_out/prof_bench -- python3 bench.py --steps 20 --warmup 20 --cpu-evals 0 --extras 0 --repeats 5 > gpurun_out/rocprof_bench.log 2>&1; rc=$?; echo "rocprof bench rc=$rc"; [ $rc -eq 0 ] || stop "rocprof bench" $rc
  grep "^{" gpurun_out/rocprof_bench.log | cut -c1-140
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-evals 0 --extras 0 > gpurun_out/torchrun_world1_nccl.log 2>&1; rc=$?; echo "torchrun world 1 rc=$rc"; [ $rc -eq 0 ] || stop "torchrun" $rc
  grep "^{" gpurun_out/torchrun_world1_nccl.log | cut -c1-140
  timeout -k 10 300 python bench.py --inproc 1 --steps 20 --warmup 5 > gpurun_out/inproc1.log 2> gpurun_out/inproc1.err; rc=$?; echo "inproc 1 rc=$rc"; [ $rc -eq 0 ] || stop "inproc 1" $rc
  grep "^{" gpurun_out/inproc1.log | cut -c1-140
  timeout -k 10 300 python bench.py --inproc 2 --devices 0,0 --steps 20 --warmup 5 --repeats 5 > gpurun_out/inproc2_one_gpu.log 2> gpurun_out/inproc2.err; rc=$?; echo "inproc 2 (one GPU) rc=$rc"; [ $rc -eq 0 ] || stop "inproc 2" $rc
  grep "^{" gpurun_out/inproc2_one_gpu.log | cut -c1-140
  timeout -k 10 150 $Q -d gpurun_out/tw3 -- python3 tools/prof_wave.py 3 8 4 > gpurun_out/tw3.log 2>&1; rc=$?; [ $rc -eq 0 ] || stop "trace G=3" $rc
  python tools/trace_wave.py gpurun_out/tw3 600 > gpurun_out/wave_trace_3x7.txt
  timeout -k 10 200 python tools/gpu_wave_profile.py 20 > gpurun_out/wave_profile.log 2>&1; rc=$?; [ $rc -eq 0 ] || stop "wave profile" $rc
  timeout -k 10 200 python tools/gpu_grad_batch.py > gpurun_out/grad_batch.log 2>&1; rc=$?; [ $rc -eq 0 ] || stop "grad batch" $rc
  timeout -k 10 200 python tools/gpu_chain_timeline.py 8192 > /dev/null 2>&1; rc=$?; [ $rc -eq 0 ] || stop "timeline" $rc
fi
if [ "$PART" != 1 ]; then
  rm -rf gpurun_out/pmc_wave1 gpurun_out/pmc_wave2 gpurun_out/pmc_wave3 gpurun_out/pmc_wave4 gpurun_out/pmc_gemm1 gpurun_out/pmc_gemm2 gpurun_out/pmc_gemm3 gpurun_out/prof_single gpurun_out/prof_predict gpurun_out/pmc_med1 gpurun_out/pmc_med2 gpurun_out/pmc_med3 gpurun_out/pmc_med4 gpurun_out/prof_medium
  W="python3 tools/prof_wave.py 3 8 4"
  timeout -k 10 200 $Q --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES -d gpurun_out/pmc_wave1 -- $W > gpurun_out/pmc_wave1.log 2>&1; rc=$?; echo "pmc wave1 rc=$rc"; [ $rc -eq 0 ] || stop "pmc wave1" $rc
  timeout -k 10 200 $Q --pmc FETCH_SIZE -d gpurun_out/pmc_wave3 -- $W > gpurun_out/pmc_wave3.log 2>&1; rc=$?; echo "pmc wave3 rc=$rc"; [ $rc -eq 0 ] || stop "pmc wave3" $rc
  timeout -k 10 200 $Q --pmc WRITE_SIZE -d gpurun_out/pmc_wave4 -- $W > gpurun_out/pmc_wave4.log 2>&1; rc=$?; echo "pmc wave4 rc=$rc"; [ $rc -eq 0 ] || stop "pmc wave4" $rc
  P="python3 tools/prof_gemm.py 7 8192 256 1 3"
  timeout -k 10 200 $Q --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES -d gpurun_out/pmc_gemm1 -- $P > gpurun_out/pmc_gemm1.log 2>&1; rc=$?; echo "pmc gemm1 rc=$rc"; [ $rc -eq 0 ] || stop "pmc gemm1" $rc
  timeout -k 10 200 $Q --pmc FETCH_SIZE -d gpurun_out/pmc_gemm2 -- $P > gpurun_out/pmc_gemm2.log 2>&1; rc=$?; echo "pmc gemm2 rc=$rc"; [ $rc -eq 0 ] || stop "pmc gemm2" $rc
  timeout -k 10 200 $Q --pmc WRITE_SIZE -d gpurun_out/pmc_gemm3 -- $P > gpurun_out/pmc_gemm3.log 2>&1; rc=$?; echo "pmc gemm3 rc=$rc"; [ $rc -eq 0 ] || stop "pmc gemm3" $rc
  M="python3 tools/prof_medium.py 2048 512 2"
  timeout -k 10 200 $R -d gpurun_out/prof_medium -- $M > gpurun_out/rocprof_medium.log 2>&1; rc=$?; echo "medium stats rc=$rc"; [ $rc -eq 0 ] || stop "medium stats" $rc
  timeout -k 10 200 $Q --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES -d gpurun_out/pmc_med1 -- $M > gpurun_out/pmc_med1.log 2>&1; rc=$?; echo "pmc med1 rc=$rc"; [ $rc -eq 0 ] || stop "pmc med1" $rc
  timeout -k 10 200 $Q --pmc FETCH_SIZE -d gpurun_out/pmc_med2 -- $M > gpurun_out/pmc_med2.log 2>&1; rc=$?; echo "pmc med2 rc=$rc"; [ $rc -eq 0 ] || stop "pmc med2" $rc
  timeout -k 10 200 $Q --pmc WRITE_SIZE -d gpurun_out/pmc_med3 -- $M > gpurun_out/pmc_med3.log 2>&1; rc=$?; echo "pmc med3 rc=$rc"; [ $rc -eq 0 ] || stop "pmc med3" $rc
  timeout -k 10 200 $Q --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 -d gpurun_out/pmc_med4 -- $M > gpurun_out/pmc_med4.log 2>&1; rc=$?; echo "pmc med4 rc=$rc"
  timeout -k 10 200 $R -d gpurun_out/prof_single -- python3 tools/prof_eval.py 2048 8192 > gpurun_out/rocprof_single.log 2>&1; rc=$?; echo "single rc=$rc"; [ $rc -eq 0 ] || stop "single" $rc
  timeout -k 10 200 $R -d gpurun_out/prof_predict -- python3 bench.py --config predict > gpurun_out/rocprof_predict.log 2>&1; rc=$?; echo "predict rc=$rc"; [ $rc -eq 0 ] || stop "predict" $rc
fi
find gpurun_out -name "*kernel_stats.csv" | head; du -sh gpurun_out
