#!/bin/bash
# round 5, final binary: soak against the oracle (whole state compared), fresh-context hunt of the final state's pinned copy
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "== random 400"; SOAK_SEED=70701 timeout 2400 python tools/gpu_soak.py 400 4000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm" | tail -3
echo "== degenerate 200"; SOAK_DEGENERATE=1 SOAK_SEED=70702 timeout 1800 python tools/gpu_soak.py 200 3000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm" | tail -3
echo "== larger clouds 60 x 12000"; SOAK_SEED=70703 timeout 2400 python tools/gpu_soak.py 60 12000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm" | tail -3
echo "== fresh contexts"; timeout 900 python tools/gpu_fresh_hunt.py 3000 32 60 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm" | tail -4
timeout 900 python tools/gpu_flaky_hunt.py 10000 64 6 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm" | tail -4
