#!/bin/bash
# A/B on the GPU box: single-stream and batched figures under a few environment settings
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOTDIR
run() { echo "== $*"; env "$@" python tools/gpu_batch.py 10000 10 1,32 2>&1 | grep "^B"; env "$@" DISTINCT=1 python tools/gpu_batch.py 10000 5 32 2>&1 | grep "^B" | sed 's/^/distinct /'; }
run CVO_HIP_GRAPH=1 CVO_HIP_MERGED_WAVES=4
run CVO_HIP_GRAPH=1 CVO_HIP_MERGED_WAVES=6
echo "== acvo"; CVO_HIP_GRAPH=1 CVO_HIP_MERGED_WAVES=4 python tools/gpu_batch.py 10000 10 1,32 acvo | grep "^B"
CVO_HIP_GRAPH=1 CVO_HIP_MERGED_WAVES=6 python tools/gpu_batch.py 10000 10 1 acvo | grep "^B"
echo "== 20k x 8"; CVO_HIP_GRAPH=1 DISTINCT=1 python tools/gpu_batch.py 20000 5 8 | grep "^B"
echo "== 3k"; CVO_HIP_GRAPH=1 python tools/gpu_batch.py 3000 10 1,32 | grep "^B"
