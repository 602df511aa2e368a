#!/bin/bash
# first GPU pass: building-block tests -> probes -> full parity suite -> bench.  A step that times out
# (124/137) stops the chain; an ordinary test failure does not.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
step() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/run1.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/run1.log
  tail -n 25 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name, stopping"; exit 1; fi
  return 0
}
step t_blocks 420 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or kernel_matrix or potrf"
step probe 300 python tools/gpu_probe.py
step t_all 900 python -m pytest tests/test_gpu_parity.py -q -m gpu
step bench 400 python bench.py --steps 10 --warmup 2
