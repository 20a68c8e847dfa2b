#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_$i.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/pytest_gpu_$i.log | tail -2
done
