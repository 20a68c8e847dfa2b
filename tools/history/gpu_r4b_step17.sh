#!/bin/bash
# probe: engines on streams of descending priority
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python - <<'PY'
import torch
print(torch.cuda.get_device_name(0))
PY
timeout 1500 python tools/gpu_abx.py 3 "X=0" "CVO_HIP_ENGINE_PRIO=1" -- "10000 6 64" "10000 3 256" "10000 6 32" 2>&1 | tee gpurun_out/r4b_ab_prio.txt
CVO_HIP_ENGINE_PRIO=1 CVO_HIP_ENGINE_DEBUG=1 DISTINCT=1 CVO_HIP_GRAPH=1 python tools/gpu_batch.py 10000 3 64 2>&1 | grep "idle after\|align_many" | tail -8
