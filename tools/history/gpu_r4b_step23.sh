#!/bin/bash
# the filter launch's pre-transform requested before the filter's first round trip
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
timeout 1500 python tools/gpu_abx_libs.py 4 libcvo_hip.so libcvo_hip_pretfafter.so -- "10000 6 64" "10000 3 256" "3000 6 64" 2>&1 | tee gpurun_out/r4b_ab_pretf.txt
for lib in libcvo_hip.so libcvo_hip_pretfafter.so; do DISTINCT=1 CVO_HIP_GRAPH=1 CVO_LIB=$lib python tools/gpu_batch.py 10000 4 64 acvo | tail -1; done
