#!/bin/bash
# GPU pass 2: parity suite with the numpy-exact exp, clock-aware probes, rocprofv3 kernel traces.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
step() {
  local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/run2.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/run2.log
  tail -n 30 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name, stopping"; exit 1; fi
  return 0
}
python -m gsum_amd.build && step t_all 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "potrf or golden or large"
step probe 300 python tools/gpu_probe.py
rm -rf gpurun_out/prof_r1
step rocprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1 -- python3 tools/prof_eval.py 2048 8192
ls -R gpurun_out/prof_r1 | head -30
