#!/bin/bash
# round 4 (second session), third GPU call: list-pass loops with pinned scalars and a prefetched next record
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -4
timeout 1500 python tools/gpu_abx_libs.py 3 libcvo_hip.so libcvo_hip_nopf.so libcvo_hip_nopin.so libcvo_hip_base0.so -- "10000 6 64" "10000 3 256" "20000 4 8" 2>&1 | tee gpurun_out/r4b_ab_pin.txt
for lib in libcvo_hip.so libcvo_hip_nopf.so libcvo_hip_nopin.so; do for cfg in "10000 40 cvo" "3000 60 cvo" "10000 30 acvo"; do echo -n "$lib: "; CVO_LIB=$lib python tools/gpu_single.py $cfg 2>&1 | grep single; done; done | tee gpurun_out/r4b_single_pin.txt
DISTINCT=1 CVO_HIP_GRAPH=1 python tools/gpu_batch.py 10000 4 64 acvo | tail -1
DISTINCT=1 CVO_HIP_GRAPH=1 CVO_LIB=libcvo_hip_nopin.so python tools/gpu_batch.py 10000 4 64 acvo | tail -1
