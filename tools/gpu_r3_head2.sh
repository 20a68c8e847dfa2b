#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CVO_HIP_GRAPH=1
run() { echo "== $*"; for n in 10000 3000; do env "$@" timeout 120 python tools/gpu_single.py $n 40 cvo 2>&1 | grep "^single\|cvo_hip"; done; }
run X=1
run CVO_HIP_HEAD_PREFETCH=1
run X=1
run CVO_HIP_HEAD_PREFETCH=1
run CVO_HIP_POST_DEBUG=1
run CVO_HIP_POST_DEBUG=1 CVO_HIP_HEAD_PREFETCH=1
run CVO_HIP_POST_DEBUG=1 CVO_HIP_NO_HEAD=1
echo "== acvo"; for n in 10000 3000; do timeout 120 python tools/gpu_single.py $n 30 acvo 2>&1 | grep "^single"; CVO_HIP_HEAD_PREFETCH=1 timeout 120 python tools/gpu_single.py $n 30 acvo 2>&1 | grep "^single"; done
