#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
echo "== 14000 default"; DISTINCT=1 timeout 300 python tools/gpu_batch.py 14000 6 2,3 2>&1 | grep "^B " | cut -c1-70
echo "== 14000 no engines"; CVO_HIP_NO_FUSE=1 DISTINCT=1 timeout 300 python tools/gpu_batch.py 14000 6 2,3 2>&1 | grep "^B " | cut -c1-70
echo "== 12000 default"; DISTINCT=1 timeout 300 python tools/gpu_batch.py 12000 6 2,3 2>&1 | grep "^B " | cut -c1-70
echo "== 12000 no engines"; CVO_HIP_NO_FUSE=1 DISTINCT=1 timeout 300 python tools/gpu_batch.py 12000 6 2,3 2>&1 | grep "^B " | cut -c1-70
