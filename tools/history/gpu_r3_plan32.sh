#!/bin/bash
# the plan in float32: parity, soaks (the degenerate one tests the bounds), one registration at a time, batched
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python -m pytest tests/test_gpu_paths.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
SOAK_SEED=4711 timeout 900 python tools/gpu_soak.py 800 4000 2>&1 | tail -1
SOAK_DEGENERATE=1 SOAK_SEED=99 timeout 1500 python tools/gpu_soak.py 1200 3000 2>&1 | tail -1
SOAK_SEED=5 timeout 1500 python tools/gpu_soak.py 150 12000 2>&1 | tail -1
for r in 1 2; do
for cfg in "3000 60 cvo" "6000 40 cvo" "10000 40 cvo" "14000 30 cvo" "3000 60 acvo" "10000 40 acvo"; do python tools/gpu_single.py $cfg 2>&1 | grep single; done
done
DISTINCT=1 CVO_HIP_GRAPH=1 python tools/gpu_batch.py 10000 6 64 2>&1 | grep "registrations/s" | tail -1
CVO_HIP_POST_DEBUG=1 python tools/gpu_single.py 10000 20 cvo 2>&1 | grep -i "post-step part" | tail -2
