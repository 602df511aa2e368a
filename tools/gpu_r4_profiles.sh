#!/bin/bash
# round-4 evidence: the bench line on the driver's command, the same under rocprofv3 --kernel-trace --stats, under torch.distributed.run
# (world 1, nccl) and as two ranks on this one GPU (gloo); per-dispatch traces of a batch call; PMC passes (one counter group per run,
# --kernel-trace only) of the batch's launches and of the one-product bulk tile.    tools/gpu_r4_profiles.sh [1|2|all]
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
  rm -rf gpurun_out/prof_bench gpurun_out/tw1 gpurun_out/tw3
  timeout -k 10 300 $R -d gpurun_out/prof_bench -- python3 bench.py --steps 20 --warmup 20 --cpu-evals 0 --extras 0 --repeats 5 > gpurun_out/rocprof_bench.log 2>&1; rc=$?; echo "rocprof bench rc=$rc"; [ $rc -eq 0 ] || stop "rocprof bench" $rc
  grep "^{" gpurun_out/rocprof_bench.log | cut -c1-140
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-evals 0 --extras 0 > gpurun_out/torchrun_world1_nccl.log 2>&1; rc=$?; echo "torchrun world 1 rc=$rc"; [ $rc -eq 0 ] || stop "torchrun" $rc
  grep "^{" gpurun_out/torchrun_world1_nccl.log | cut -c1-140
  timeout -k 10 300 python bench.py --gpus 2 --backend gloo --device 0 --steps 20 --warmup 3 --cpu-evals 0 --extras 0 > gpurun_out/two_ranks_one_gpu_gloo.log 2>&1; rc=$?; echo "two ranks gloo rc=$rc"; [ $rc -eq 0 ] || stop "two ranks" $rc
  grep "^{" gpurun_out/two_ranks_one_gpu_gloo.log | cut -c1-140
  timeout -k 10 150 $Q -d gpurun_out/tw1 -- python3 tools/prof_wave.py 1 20 4 > gpurun_out/tw1.log 2>&1; rc=$?; [ $rc -eq 0 ] || stop "trace G=1" $rc
  timeout -k 10 150 $Q -d gpurun_out/tw3 -- python3 tools/prof_wave.py 3 8 4 > gpurun_out/tw3.log 2>&1; rc=$?; [ $rc -eq 0 ] || stop "trace G=3" $rc
  python tools/trace_wave.py gpurun_out/tw1 400 > gpurun_out/wave_trace_1x20.txt; python tools/trace_wave.py gpurun_out/tw3 600 > gpurun_out/wave_trace_3x7.txt
  tail -12 gpurun_out/wave_trace_3x7.txt
  timeout -k 10 200 python tools/gpu_wave_profile.py 20 > gpurun_out/wave_profile.log 2>&1; rc=$?; [ $rc -eq 0 ] || stop "wave profile" $rc
  rm -f gpurun_out/chain_abort_repro.log; for v in fresh bench b3 b20 torch_nccl_b20; do timeout -k 10 120 python tools/gpu_chain_abort_repro.py $v 2>&1 | grep "^{" >> gpurun_out/chain_abort_repro.log || stop "single after batch ($v)" 1; done
  (cd tools && timeout -k 10 400 python gpu_wave_sweep.py K=20 K=20,wave_groups=2,wave_size=10 K=20,wave_groups=4,wave_size=5 K=20,wave_depth=2 K=20,wave_depth=3 K=20,wave_depth=6 K=20,wave_panel_wg4=0 K=20,wave_panel_wg4=8 K=20,wave_head=0 K=20,wave_serial=1 K=64,reps=3 K=64,reps=3,wave_size=7 K=84,reps=3 K=84,reps=3,wave_shift=4 K=20 > ../gpurun_out/wave_sweep.log 2>&1); rc=$?; [ $rc -eq 0 ] || stop "wave sweep" $rc
fi
if [ "$PART" != 1 ]; then
  rm -rf gpurun_out/pmc_wave1 gpurun_out/pmc_wave2 gpurun_out/pmc_wave3 gpurun_out/pmc_wave4 gpurun_out/pmc_gemm1 gpurun_out/pmc_gemm2 gpurun_out/pmc_gemm3 gpurun_out/prof_single gpurun_out/prof_predict
  W="python3 tools/prof_wave.py 3 8 4"
  timeout -k 10 200 $Q --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES -d gpurun_out/pmc_wave1 -- $W > gpurun_out/pmc_wave1.log 2>&1; rc=$?; echo "pmc wave1 rc=$rc"; [ $rc -eq 0 ] || stop "pmc wave1" $rc
  timeout -k 10 200 $Q --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d gpurun_out/pmc_wave2 -- $W > gpurun_out/pmc_wave2.log 2>&1; rc=$?; echo "pmc wave2 rc=$rc"; [ $rc -eq 0 ] || stop "pmc wave2" $rc
  timeout -k 10 200 $Q --pmc FETCH_SIZE -d gpurun_out/pmc_wave3 -- $W > gpurun_out/pmc_wave3.log 2>&1; rc=$?; echo "pmc wave3 rc=$rc"; [ $rc -eq 0 ] || stop "pmc wave3" $rc
  timeout -k 10 200 $Q --pmc WRITE_SIZE -d gpurun_out/pmc_wave4 -- $W > gpurun_out/pmc_wave4.log 2>&1; rc=$?; echo "pmc wave4 rc=$rc"; [ $rc -eq 0 ] || stop "pmc wave4" $rc
  P="python3 tools/prof_gemm.py 7 8192 256 1 3"
  timeout -k 10 200 $Q --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES -d gpurun_out/pmc_gemm1 -- $P > gpurun_out/pmc_gemm1.log 2>&1; rc=$?; echo "pmc gemm1 rc=$rc"; [ $rc -eq 0 ] || stop "pmc gemm1" $rc
  timeout -k 10 200 $Q --pmc FETCH_SIZE -d gpurun_out/pmc_gemm2 -- $P > gpurun_out/pmc_gemm2.log 2>&1; rc=$?; echo "pmc gemm2 rc=$rc"; [ $rc -eq 0 ] || stop "pmc gemm2" $rc
  timeout -k 10 200 $Q --pmc WRITE_SIZE -d gpurun_out/pmc_gemm3 -- $P > gpurun_out/pmc_gemm3.log 2>&1; rc=$?; echo "pmc gemm3 rc=$rc"; [ $rc -eq 0 ] || stop "pmc gemm3" $rc
  timeout -k 10 200 $R -d gpurun_out/prof_single -- python3 tools/prof_eval.py 2048 8192 > gpurun_out/rocprof_single.log 2>&1; rc=$?; echo "single rc=$rc"; [ $rc -eq 0 ] || stop "single" $rc
  timeout -k 10 200 $R -d gpurun_out/prof_predict -- python3 bench.py --config predict > gpurun_out/rocprof_predict.log 2>&1; rc=$?; echo "predict rc=$rc"; [ $rc -eq 0 ] || stop "predict" $rc
fi
find gpurun_out -name "*kernel_stats.csv" | head; du -sh gpurun_out
