#!/bin/bash
# round-3 soak on the final binary: head mode (default) against the oracle
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "== random 2000"; SOAK_SEED=31337 timeout 2400 python tools/gpu_soak.py 2000 4000 2>&1 | tail -2
echo "== degenerate 800"; SOAK_DEGENERATE=1 SOAK_SEED=778 timeout 1800 python tools/gpu_soak.py 800 3000 2>&1 | tail -2
echo "== larger clouds 300 x 12000"; SOAK_SEED=4243 timeout 2400 python tools/gpu_soak.py 300 12000 2>&1 | tail -2
echo "== graphs on, 600"; CVO_HIP_GRAPH=1 SOAK_SEED=98 timeout 1800 python tools/gpu_soak.py 600 3500 2>&1 | tail -2
echo "== MATLAB weight 300"; SOAK_MATLAB=1 SOAK_SEED=6 timeout 900 python tools/gpu_soak.py 300 2500 2>&1 | tail -2
echo "== front end 300"; timeout 900 python tools/gpu_soak_fe.py 300 2>&1 | tail -1
