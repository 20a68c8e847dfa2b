#!/bin/bash
# engine count on the pipelined kernels
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python tools/gpu_abx.py 2 "X=0" "CVO_HIP_ENGINES_FORCE=2" "CVO_HIP_ENGINES_FORCE=3" "CVO_HIP_ENGINES_FORCE=4" -- "10000 6 64" "10000 3 256" "10000 6 32" 2>&1 | tee gpurun_out/r4b_ab_engines.txt
