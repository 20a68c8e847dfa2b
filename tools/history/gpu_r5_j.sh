#!/bin/bash
# the last registrations of an align_many call leave their engines for launches of their own: parity subset, then `value` with the switch at 0 (off) ... 4
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -m gpu -x -q -k "align_many or mixed_bag or refills or headline or resident or engine" > gpurun_out/pytest_gpu.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/pytest_gpu.log | tail -4
for at in 0 3 0 3 1 2 4; do
  echo "-- CVO_HIP_MIGRATE_AT=$at"
  CVO_HIP_MIGRATE_AT=$at DISTINCT=1 timeout 300 python tools/gpu_batch.py 10000 8 64,256 2>&1 | grep "^B \|mismatch\|Error\|error" | head -6
done
