#!/bin/bash
# a registration's share of the compute units grows as the call's others end ("share_grows"), off / on, interleaved
cd ${GRAFT_REPO_ROOT:-.}
export DISTINCT=1 CVO_HIP_GRAPH=1
for round in 1 2 3; do
  echo "== off"; CVO_HIP_NO_SHARE_GROW=1 python tools/gpu_batch.py 10000 8 64,256 2>&1 | grep "^B" | cut -c1-60
  echo "== on";  python tools/gpu_batch.py 10000 8 64,256 2>&1 | grep "^B" | cut -c1-60
done
for v in off on; do
  [ $v = off ] && export CVO_HIP_NO_SHARE_GROW=1 || unset CVO_HIP_NO_SHARE_GROW
  echo "== 3k $v"; python tools/gpu_batch.py 3000 8 2,4,8,32,64 2>&1 | grep "^B" | cut -c1-60
  echo "== acvo $v"; python tools/gpu_batch.py 10000 6 64 acvo 2>&1 | grep "^B" | cut -c1-60
  echo "== 10k small $v"; python tools/gpu_batch.py 10000 8 2,4,8 2>&1 | grep "^B" | cut -c1-60
done
