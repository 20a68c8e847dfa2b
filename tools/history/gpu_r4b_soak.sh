#!/bin/bash
# round 4, second session: soak of the re-arranged list passes / filter loop against the oracle (new seeds)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "== random 1000"; SOAK_SEED=50501 timeout 2400 python tools/gpu_soak.py 1000 4000 2>&1 | tail -2
echo "== degenerate 400"; SOAK_DEGENERATE=1 SOAK_SEED=50502 timeout 1800 python tools/gpu_soak.py 400 3000 2>&1 | tail -2
echo "== larger clouds 120 x 12000"; SOAK_SEED=50503 timeout 2400 python tools/gpu_soak.py 120 12000 2>&1 | tail -2
echo "== graphs on, 300"; CVO_HIP_GRAPH=1 SOAK_SEED=50504 timeout 1800 python tools/gpu_soak.py 300 3500 2>&1 | tail -2
echo "== MATLAB weight 150"; SOAK_MATLAB=1 SOAK_SEED=50505 timeout 900 python tools/gpu_soak.py 150 2500 2>&1 | tail -2
echo "== above 65536 rows"; timeout 2400 python tools/gpu_soak_big.py 2>&1 | tail -3
