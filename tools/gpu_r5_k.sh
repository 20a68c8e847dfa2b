#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/pytest_gpu.log | tail -5
echo "== random 300"; SOAK_SEED=80801 timeout 2400 python tools/gpu_soak.py 300 4000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm" | tail -2
echo "== larger clouds 80 x 12000"; SOAK_SEED=80803 timeout 2400 python tools/gpu_soak.py 80 12000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm" | tail -2
echo "== big clouds 30 x 16000"; SOAK_SEED=80804 timeout 2400 python tools/gpu_soak.py 30 16000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm" | tail -2
