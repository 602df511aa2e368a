#!/bin/bash
# after a diag-kernel change: parity tests, probe (stage timings, batch sweep, diag stamps), default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m gsum_amd.build
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/gpu_probe.py > gpurun_out/probe.log 2>&1; echo rc=$?
grep "batch_\|^8192 1 1\|stamps" gpurun_out/probe.log | tail -14
timeout -k 10 300 python bench.py --cpu-evals 0 2>&1 | tail -1 | cut -c1-900
