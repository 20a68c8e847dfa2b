#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/pytest_gpu.log | tail -4
for n in 3000 6000; do
DISTINCT=1 timeout 300 python tools/gpu_batch.py $n 8 2,4,8,16 acvo 2>&1 | grep "^B " | cut -c1-70
done
echo "== acvo 10000: default, then CVO_HIP_NO_FUSE"
DISTINCT=1 timeout 300 python tools/gpu_batch.py 10000 8 2,3,4 acvo 2>&1 | grep "^B " | cut -c1-70
CVO_HIP_NO_FUSE=1 DISTINCT=1 timeout 300 python tools/gpu_batch.py 10000 8 2,3,4 acvo 2>&1 | grep "^B " | cut -c1-70
