#!/bin/bash
# round-1 evidence run: full parity suite, smoke, bench (with CPU baseline), rocprofv3 stats of the bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m gsum_amd.build
step() {
  local name=$1 to=$2; shift 2
  echo "=== $name"
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc"
  tail -n 12 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name, stopping"; exit 1; fi
  return 0
}
step t_all 900 python -m pytest tests -q -m gpu
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step bench 600 python bench.py
rm -rf gpurun_out/prof_bench
step rocprof_bench 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 16 --warmup 4 --cpu-evals 0
find gpurun_out/prof_bench -name "*stats*"
