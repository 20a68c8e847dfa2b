#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python tools/gpu_r5_run_check.py 3000 6000 10000 2>&1 | grep -v amdgpu.ids


for n in 10000 3000; do python tools/gpu_single_phases.py $n 30 cvo 2>&1 | grep -v amdgpu.ids; done
CVO_LIB=libcvo_hip_clk.so python tools/gpu_run_clocks.py 3000 10000 2>&1 | grep -v amdgpu.ids
