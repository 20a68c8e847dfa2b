#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
for n in 10000 3000; do DISTINCT=1 timeout 300 python tools/gpu_batch.py $n 8 1 2>&1 | grep "^B " ; done
DISTINCT=1 timeout 300 python tools/gpu_batch.py 10000 8 64 2>&1 | grep "^B "
