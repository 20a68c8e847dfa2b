#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python tools/gpu_abx_libs.py 5 libcvo_hip.so libcvo_hip_noepf.so -- "10000 6 64" "10000 4 256" 2>&1 | tee gpurun_out/r4b_ab_epf2.txt
