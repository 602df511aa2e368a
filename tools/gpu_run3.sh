#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m gsum_amd.build
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 9 --warmup 3 --cpu-evals 0 2>&1 | tail -4 | cut -c1-600
