#!/bin/bash
# Taylor constants by a wave instead of one lane (kt_step_twist, kt_post_flow)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -4
for r in 1 2; do for lib in libcvo_hip.so libcvo_hip_serialxi.so; do for cfg in "10000 40 cvo" "3000 60 cvo" "10000 30 acvo" "3000 40 acvo"; do echo -n "$lib: "; CVO_LIB=$lib python tools/gpu_single.py $cfg 2>&1 | grep single; done; done; done | tee gpurun_out/r4b_single_xi.txt
timeout 1500 python tools/gpu_abx_libs.py 3 libcvo_hip.so libcvo_hip_serialxi.so -- "10000 6 64" "10000 3 256" 2>&1 | tee gpurun_out/r4b_ab_xi.txt
