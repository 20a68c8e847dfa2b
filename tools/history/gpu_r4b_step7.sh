#!/bin/bash
# waves per SIMD of the engines' flow pass on the pipelined loops
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python tools/gpu_abx_libs.py 3 libcvo_hip.so libcvo_hip_w6.so libcvo_hip_w5.so -- "10000 6 64" "10000 3 256" "20000 4 8" 2>&1 | tee gpurun_out/r4b_ab_waves.txt
