#!/bin/bash
# round-4 soak on the final binary against the oracle (new seeds), plus the hand-over soak
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "== hand-over 60 x 6"; timeout 900 python tools/gpu_soak_handover.py 60 6 2>&1 | tail -3
echo "== random 1200"; SOAK_SEED=40401 timeout 2400 python tools/gpu_soak.py 1200 4000 2>&1 | tail -2
echo "== degenerate 500"; SOAK_DEGENERATE=1 SOAK_SEED=40402 timeout 1800 python tools/gpu_soak.py 500 3000 2>&1 | tail -2
echo "== larger clouds 150 x 12000"; SOAK_SEED=40403 timeout 2400 python tools/gpu_soak.py 150 12000 2>&1 | tail -2
echo "== graphs on, 400"; CVO_HIP_GRAPH=1 SOAK_SEED=40404 timeout 1800 python tools/gpu_soak.py 400 3500 2>&1 | tail -2
echo "== MATLAB weight 200"; SOAK_MATLAB=1 SOAK_SEED=40405 timeout 900 python tools/gpu_soak.py 200 2500 2>&1 | tail -2
echo "== front end 200"; SOAK_SEED=40406 timeout 900 python tools/gpu_soak_fe.py 200 2>&1 | tail -1
