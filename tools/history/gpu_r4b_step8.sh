#!/bin/bash
# expansion passes one batch behind their gathers
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -4
timeout 1500 python tools/gpu_abx_libs.py 3 libcvo_hip.so libcvo_hip_w4.so libcvo_hip_noxp.so -- "10000 6 64" "10000 3 256" "20000 4 8" "3000 6 64" 2>&1 | tee gpurun_out/r4b_ab_xpipe.txt
for lib in libcvo_hip.so libcvo_hip_noxp.so; do for cfg in "10000 40 cvo" "200000 3 cvo" "70000 5 cvo"; do echo -n "$lib: "; CVO_LIB=$lib python tools/gpu_single.py $cfg 2>&1 | grep single; done; done | tee gpurun_out/r4b_single_xpipe.txt
for lib in libcvo_hip.so libcvo_hip_noxp.so; do DISTINCT=1 CVO_HIP_GRAPH=1 CVO_LIB=$lib python tools/gpu_batch.py 10000 4 64 acvo | tail -1; done
