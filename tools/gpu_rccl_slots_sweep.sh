#!/bin/bash
# Evaluations in flight vs throughput with an RCCL communicator in the process (torchrun, world 1, nccl) and without:
# the measurement behind the default of 16 (20 collapse to 148 evals/s under RCCL; DESIGN.md section 4, batch pipelining).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
A="--cpu-evals 0 --extras 0 --repeats 3 --warmup 3 --steps 32"
show() { grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'evals/s', d['queue_probe'], [round(v) for v in d['repeats']['evals_per_s_all']])"; }
{
for s in 8 12 14 16 20; do
  echo "torchrun nccl, slots=$s:"; timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 $A --backend nccl --slots $s 2>&1 | show
done
for s in 16 20; do echo "plain process, slots=$s:"; timeout -k 10 200 python bench.py $A --slots $s 2>&1 | show; done
} | tee gpurun_out/slots_sweep.log
