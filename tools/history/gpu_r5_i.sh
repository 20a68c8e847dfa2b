#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5
SOAK_SEED=70701 timeout 1200 python tools/gpu_soak.py 400 6000 2>&1 | tail -1
SOAK_DEGENERATE=1 SOAK_SEED=70702 timeout 1200 python tools/gpu_soak.py 240 3000 2>&1 | tail -1
SOAK_SEED=70703 timeout 1200 python tools/gpu_soak.py 80 14000 2>&1 | tail -1
python tools/gpu_stream.py 2>&1 | grep "cvo prefetch" | tail -2
