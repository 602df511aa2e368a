#!/bin/bash
# What clock and power the chip runs at under the bulk kernel: rocm-smi sampled while tools/prof_gemm.py loops the M = 8192,
# K = 256 SYRK launch (2000 launches ~ 0.7 s per run), and idle.  Evidence for how far the 2.4 GHz peak is from the
# sustained fp64-MFMA clock.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "--- idle"; rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power\|fclk\|mclk" | head -8
python3 tools/prof_gemm.py 7 8192 256 1 6000 > gpurun_out/clock_gemm.log 2>&1 &
PID=$!
sleep 1.0
for i in 1 2 3 4; do echo "--- under load, sample $i"; rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power" | head -4; sleep 0.3; done
wait $PID
cat gpurun_out/clock_gemm.log | tail -2
rocm-smi --showmaxpower 2>&1 | grep -i "power" | head -3
