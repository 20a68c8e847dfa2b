#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for n in 3000 10000; do
echo "== acvo $n engines"
ACVO=1 DISTINCT=1 timeout 300 python tools/gpu_batch.py $n 8 2,4,8,16 2>&1 | grep "^B " | cut -c1-70
echo "== acvo $n no engines"
ACVO=1 CVO_HIP_NO_FUSE=1 DISTINCT=1 timeout 300 python tools/gpu_batch.py $n 8 2,4,8,16 2>&1 | grep "^B " | cut -c1-70
done
