#!/bin/bash
# engines, once the whole call is narrow: blocks of a list pass per registration ("narrow_blocks": 0 = as ever, 64 at 24 slots), the twist in the step launch ("narrow_merge")
cd ${GRAFT_REPO_ROOT:-.}
export DISTINCT=1 CVO_HIP_GRAPH=1
for round in 1 2; do
  for nb in 0 32 16 8; do
    echo "== narrow_blocks $nb"; CVO_HIP_NARROW_BLOCKS=$nb python tools/gpu_batch.py 10000 8 64,256 2>&1 | grep "^B" | cut -c1-60
  done
done
for nb in 0 16; do
echo "== 3k narrow_blocks $nb"; CVO_HIP_NARROW_BLOCKS=$nb python tools/gpu_batch.py 3000 8 32,64 2>&1 | grep "^B" | cut -c1-60
echo "== acvo narrow_blocks $nb"; CVO_HIP_NARROW_BLOCKS=$nb python tools/gpu_batch.py 10000 6 64 acvo 2>&1 | grep "^B" | cut -c1-60
done
