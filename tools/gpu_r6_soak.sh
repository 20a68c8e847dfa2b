#!/bin/bash
# round 6: soak against the oracle (whole state compared) with acvo in resident runs, small align_many calls sized per call
cd ${GRAFT_REPO_ROOT:-$(pwd)}
F='^RCCL\|^HIP\|^ROCm'
echo "== random ${1:-600}"; SOAK_SEED=${SEED:-60601} timeout 1500 python tools/gpu_soak.py ${1:-600} 4000 2>&1 | grep -v "$F" | tail -3
echo "== degenerate ${2:-300}"; SOAK_DEGENERATE=1 SOAK_SEED=$((${SEED:-60601} + 1)) timeout 1200 python tools/gpu_soak.py ${2:-300} 3000 2>&1 | grep -v "$F" | tail -3
echo "== larger clouds ${3:-60} x 12000"; SOAK_SEED=$((${SEED:-60601} + 2)) timeout 1500 python tools/gpu_soak.py ${3:-60} 12000 2>&1 | grep -v "$F" | tail -3
