#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "== xcd barrier"; timeout 120 tools/microbench/xcd_barrier
export CVO_HIP_GRAPH=1 DISTINCT=1 CVO_HIP_SEGREGATE_MIN=0
run() { echo "== $*"; env "$@" timeout 300 python tools/gpu_batch.py 10000 12 64,256 2>&1 | grep "^B"; }
run A=0
run CVO_HIP_TAILS=2
run CVO_HIP_TAILS=3
run A=0
run CVO_HIP_TAILS=2
run3() { echo "== 3k $*"; env "$@" timeout 300 python tools/gpu_batch.py 3000 12 64 2>&1 | grep "^B"; }
run3 A=0
run3 CVO_HIP_TAILS=2
run3 CVO_HIP_TAILS=3
