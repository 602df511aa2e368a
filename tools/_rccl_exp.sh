cd "${GRAFT_REPO_ROOT:-/root/repo}"
A="--steps 20 --warmup 3 --cpu-evals 0 --extras 0 --repeats 3"
run() { timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 $A --backend nccl "$@" 2>&1 | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['queue_probe'], [round(v) for v in d['repeats']['evals_per_s_all']])"; }
for q in 32 24 16; do for s in 8 12 14 16 20; do echo "queues=$q slots=$s:"; GPU_MAX_HW_QUEUES=$q run --slots $s; done; done
