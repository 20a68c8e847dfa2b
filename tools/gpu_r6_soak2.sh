#!/bin/bash
# round 6, last binary: the long soak (whole state compared), captured batches, fresh contexts
cd ${GRAFT_REPO_ROOT:-$(pwd)}
F='^RCCL\|^HIP\|^ROCm\|^/opt\|^Hostname\|^Librccl'
echo "== random 1500"; SOAK_SEED=66601 timeout 3000 python tools/gpu_soak.py 1500 4000 2>&1 | grep -v "$F" | tail -2
echo "== degenerate 500"; SOAK_DEGENERATE=1 SOAK_SEED=66602 timeout 2400 python tools/gpu_soak.py 500 3000 2>&1 | grep -v "$F" | tail -2
echo "== larger clouds 150 x 12000"; SOAK_SEED=66603 timeout 2400 python tools/gpu_soak.py 150 12000 2>&1 | grep -v "$F" | tail -2
echo "== big clouds 40 x 16000"; SOAK_SEED=66604 timeout 2400 python tools/gpu_soak.py 40 16000 2>&1 | grep -v "$F" | tail -2
echo "== captured batches 200"; CVO_HIP_RUN_GRAPHS=1 CVO_HIP_GRAPH=1 SOAK_SEED=66605 timeout 2400 python tools/gpu_soak.py 200 3500 2>&1 | grep -v "$F" | tail -2
echo "== side builds 300"; CVO_HIP_SIDE=1 SOAK_SEED=66606 timeout 2400 python tools/gpu_soak.py 300 4000 2>&1 | grep -v "$F" | tail -2
echo "== fresh contexts"; timeout 900 python tools/gpu_fresh_hunt.py 3000 32 60 2>&1 | grep -v "$F" | tail -2
timeout 900 python tools/gpu_flaky_hunt.py 10000 64 8 2>&1 | grep -v "$F" | tail -2
