#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m gsum_amd.build
step() {
  local name=$1 to=$2; shift 2
  echo "=== $name"
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc"
  tail -n 40 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name, stopping"; exit 1; fi
  return 0
}
step t_all 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x
step probe 400 python tools/gpu_probe.py
step bench 400 python bench.py --steps 16 --warmup 4 --cpu-evals 0
