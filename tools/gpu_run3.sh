#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
python -m gsum_amd.build
rm -rf gpurun_out/prof_batch
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_batch -- python3 tools/prof_batch.py 8192 8 4 2>&1 | grep "ms/eval"
