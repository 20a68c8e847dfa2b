#!/bin/bash
# round 5, last binary: a longer soak against the oracle (whole state compared), fresh-context hunts at two sizes
cd ${GRAFT_REPO_ROOT:-$(pwd)}
F='^RCCL\|^HIP\|^ROCm'
echo "== random 1500"; SOAK_SEED=90901 timeout 3000 python tools/gpu_soak.py 1500 4000 2>&1 | grep -v "$F" | tail -3
echo "== degenerate 500"; SOAK_DEGENERATE=1 SOAK_SEED=90902 timeout 2400 python tools/gpu_soak.py 500 3000 2>&1 | grep -v "$F" | tail -3
echo "== larger clouds 150 x 12000"; SOAK_SEED=90903 timeout 2400 python tools/gpu_soak.py 150 12000 2>&1 | grep -v "$F" | tail -3
echo "== big clouds 40 x 16000"; SOAK_SEED=90904 timeout 2400 python tools/gpu_soak.py 40 16000 2>&1 | grep -v "$F" | tail -3
echo "== captured batches 200"; CVO_HIP_RUN_GRAPHS=1 CVO_HIP_GRAPH=1 SOAK_SEED=90905 timeout 2400 python tools/gpu_soak.py 200 3500 2>&1 | grep -v "$F" | tail -3
echo "== fresh contexts"; timeout 900 python tools/gpu_fresh_hunt.py 3000 32 60 2>&1 | grep -v "$F" | tail -3
timeout 900 python tools/gpu_fresh_hunt.py 10000 32 12 2>&1 | grep -v "$F" | tail -3
timeout 900 python tools/gpu_flaky_hunt.py 10000 64 8 2>&1 | grep -v "$F" | tail -3
