#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/pytest_gpu.log | tail -4
SEEDS=20190402 python tools/gpu_single_rate.py 3000 10000 2>&1 | grep "^n " | cut -c1-150
