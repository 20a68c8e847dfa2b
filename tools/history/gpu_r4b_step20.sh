#!/bin/bash
# expansion passes with the next 64 tile entries requested ahead (flow kernel at 6 waves per SIMD)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
timeout 1500 python tools/gpu_abx_libs.py 3 libcvo_hip.so libcvo_hip_noepf.so -- "10000 6 64" "10000 3 256" "20000 4 8" 2>&1 | tee gpurun_out/r4b_ab_epf.txt
for r in 1 2; do for lib in libcvo_hip.so libcvo_hip_noepf.so; do for cfg in "200000 3 cvo" "70000 5 cvo" "10000 40 cvo"; do echo -n "$lib: "; CVO_LIB=$lib python tools/gpu_single.py $cfg 2>&1 | grep single; done; done; done | tee gpurun_out/r4b_single_epf.txt
