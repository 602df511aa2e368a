#!/bin/bash
# round-end rehearsal: full GPU suite, smoke, bench under rocprofv3 (kernel trace + stats), plain bench, gradient profile
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; rc=$?; tail -2 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" || exit 1
rm -rf gpurun_out/prof_bench gpurun_out/prof_grad
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 18 --warmup 3 --cpu-evals 0 > gpurun_out/rocprof_bench.log 2>&1; echo "rocprof bench rc=$?"
grep "^{" gpurun_out/rocprof_bench.log | cut -c1-160
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_grad -- python3 tools/prof_grad.py 8192 3 > gpurun_out/rocprof_grad.log 2>&1; echo "rocprof grad rc=$?"
timeout -k 10 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
grep "^{" gpurun_out/bench.log | cut -c1-200
timeout -k 10 400 python tools/gpu_probe.py > gpurun_out/probe.log 2>&1; echo "probe rc=$?"
