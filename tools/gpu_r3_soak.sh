#!/bin/bash
# round-3 soak: head mode and the classic launches against the oracle
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "== random 600 (head mode default)"; timeout 1500 python tools/gpu_soak.py 600 4000 2>&1 | tail -3
echo "== degenerate 320"; SOAK_DEGENERATE=1 SOAK_SEED=777 timeout 900 python tools/gpu_soak.py 320 3000 2>&1 | tail -3
echo "== larger clouds 100 x 12000"; SOAK_SEED=4242 timeout 1500 python tools/gpu_soak.py 100 12000 2>&1 | tail -3
echo "== graphs on, 300"; CVO_HIP_GRAPH=1 SOAK_SEED=99 timeout 900 python tools/gpu_soak.py 300 3500 2>&1 | tail -3
echo "== MATLAB weight 150"; SOAK_MATLAB=1 SOAK_SEED=5 timeout 900 python tools/gpu_soak.py 150 2500 2>&1 | tail -3
echo "== front end 150"; timeout 600 python tools/gpu_soak_fe.py 150 2>&1 | tail -2
