#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m gsum_amd.build
timeout -k 10 600 python -m pytest tests -q -m gpu 2>&1 | tail -15
