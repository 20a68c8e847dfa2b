#!/bin/bash
# the streaming flow pass collects its members in LDS and writes them out 256 at a time
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -4
for r in 1 2; do for lib in libcvo_hip.so libcvo_hip_noklds.so; do for mi in 4 21; do
  echo -n "$lib max_iter $mi: "; CVO_LIB=$lib MAX_ITER=$mi DISTINCT=1 CVO_HIP_GRAPH=1 python tools/gpu_batch.py 10000 6 64 2>/dev/null | tail -1 | cut -c1-60
done; done; done 2>&1 | tee gpurun_out/r4b_klds_heavy.txt
timeout 1500 python tools/gpu_abx_libs.py 3 libcvo_hip.so libcvo_hip_noklds.so -- "10000 6 64" "10000 3 256" "20000 4 8" "3000 6 64" 2>&1 | tee gpurun_out/r4b_ab_klds.txt
for lib in libcvo_hip.so libcvo_hip_noklds.so; do DISTINCT=1 CVO_HIP_GRAPH=1 CVO_LIB=$lib python tools/gpu_batch.py 10000 4 64 acvo | tail -1; done
