#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
MAX_ITER=11 CVO_LIB=libcvo_hip_clk.so python tools/gpu_run_clocks.py 10000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm" | cut -c1-260
CVO_LIB=libcvo_hip_clk.so python tools/gpu_run_clocks.py 3000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm" | cut -c1-260
python tools/gpu_r5_run_check.py 3000 10000 2>&1 | tail -9
for i in 1 2; do SEEDS=20190402 python tools/gpu_single_rate.py 3000 6000 10000 2>&1 | grep "^n "; done
