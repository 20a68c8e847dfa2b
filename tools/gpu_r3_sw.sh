#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CVO_HIP_GRAPH=1
one() { for N in 10000 6000 3000 14000; do for M in cvo; do env "$@" timeout 120 python tools/gpu_single.py $N 30 $M 2>&1 | grep "^single" | sed "s/^/[$*] /" | cut -c1-130; done; done; for N in 10000 3000; do env "$@" timeout 120 python tools/gpu_single.py $N 30 acvo 2>&1 | grep "^single" | sed "s/^/[$*] /" | cut -c1-130; done; }
for r in 1 2; do
one X=1
one CVO_HIP_BUILD_AT=0.85
one CVO_HIP_BUILD_AT=0.9
one CVO_HIP_BUILD_AT=0.95
one CVO_HIP_BUILD_AT=0.9 CVO_HIP_LIST_MARGIN=0.3
done
