#!/bin/bash
# the streaming flow pass with its rows in LDS (tile runs of the candidate record)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -4
for r in 1 2; do for lib in libcvo_hip.so libcvo_hip_notile.so libcvo_hip_nt_singles.so; do for mi in 4 21; do
  echo -n "$lib max_iter $mi: "; CVO_LIB=$lib MAX_ITER=$mi DISTINCT=1 CVO_HIP_GRAPH=1 python tools/gpu_batch.py 10000 6 64 2>/dev/null | tail -1 | cut -c1-60
done; done; done 2>&1 | tee gpurun_out/r4b_tiled_heavy.txt
timeout 1500 python tools/gpu_abx_libs.py 3 libcvo_hip.so libcvo_hip_notile.so libcvo_hip_nt_singles.so -- "10000 6 64" "10000 3 256" "3000 6 64" 2>&1 | tee gpurun_out/r4b_ab_tiled.txt
for lib in libcvo_hip.so libcvo_hip_notile.so libcvo_hip_nt_singles.so; do DISTINCT=1 CVO_HIP_GRAPH=1 CVO_LIB=$lib python tools/gpu_batch.py 10000 4 64 acvo | tail -1; done
