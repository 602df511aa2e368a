#!/bin/bash
# RCCL on record: bench.py under torchrun at world 1 (nccl backend = RCCL), the plain N = 1 line beside it, and the
# two-ranks-on-one-GPU rehearsal of the N > 1 path through bench.py's own launcher (gloo; RCCL refuses a duplicate device).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m gsum_amd.build
A="--steps 32 --warmup 3 --cpu-evals 0 --extras 0 --repeats 5"
NCCL_DEBUG=INFO timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 $A --backend nccl > gpurun_out/rccl_world1.log 2>&1; echo "world1 nccl rc=$?"
grep -c "NCCL INFO" gpurun_out/rccl_world1.log; grep "^{" gpurun_out/rccl_world1.log | cut -c1-160
timeout -k 10 300 python bench.py $A > gpurun_out/plain_n1.log 2>&1; echo "plain rc=$?"; grep "^{" gpurun_out/plain_n1.log | cut -c1-160
timeout -k 10 300 python bench.py --gpus 2 --device 0 --backend gloo --steps 10 --warmup 2 --cpu-evals 0 --extras 0 --repeats 3 --slots 8 > gpurun_out/two_ranks_one_gpu_gloo.log 2>&1; echo "2 ranks gloo rc=$?"
grep "^{" gpurun_out/two_ranks_one_gpu_gloo.log | cut -c1-200
NCCL_DEBUG=WARN timeout -k 10 120 python bench.py --gpus 2 --device 0 --backend nccl --steps 4 --warmup 1 --cpu-evals 0 --extras 0 --repeats 1 --slots 4 > gpurun_out/two_ranks_one_gpu_nccl.log 2>&1; echo "2 ranks nccl (expected to be refused: duplicate device) rc=$?"
tail -3 gpurun_out/two_ranks_one_gpu_nccl.log | cut -c1-300
