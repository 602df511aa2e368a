cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { timeout -k 10 200 python bench.py --cpu-evals 0 --extras 0 --repeats 4 --warmup 3 "$@" 2>&1 | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), [round(v) for v in d['repeats']['evals_per_s_all']])"; }
for s in 16 20; do for k in 20 32 48; do echo "plain slots=$s steps=$k:"; run --slots $s --steps $k; done; done
echo "torchrun nccl slots=16 steps=32:"; timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --cpu-evals 0 --extras 0 --repeats 4 --warmup 3 --slots 16 --steps 32 2>&1 | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), [round(v) for v in d['repeats']['evals_per_s_all']])"
