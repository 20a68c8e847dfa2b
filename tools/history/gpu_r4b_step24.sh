#!/bin/bash
# probe: iterations per captured batch of an engine and batches kept queued, on the round-4 kernels
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 2400 python tools/gpu_abx.py 3 "X=0" "CVO_HIP_ENGINE_BATCH=6" "CVO_HIP_ENGINE_BATCH=8" "CVO_HIP_ENGINE_BATCH=14" "CVO_HIP_ENGINE_BATCH=20" "CVO_HIP_ENGINE_DEPTH=3" "CVO_HIP_ENGINE_BATCH=6 CVO_HIP_ENGINE_DEPTH=3" -- "10000 6 64" "10000 3 256" 2>&1 | tee gpurun_out/r4b_ab_batchlen.txt
