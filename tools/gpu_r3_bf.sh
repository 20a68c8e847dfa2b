#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CVO_HIP_GRAPH=1
run() { echo "== $*"; for n in 10000 3000; do env "$@" timeout 120 python tools/gpu_single.py $n 40 cvo 2>&1 | grep "^single"; done; }
for r in 1 2; do
run X=1
run CVO_HIP_BUILD_FIRST=1
run CVO_HIP_BUILD_DIV=2
run CVO_HIP_BUILD_FIRST=1 CVO_HIP_BUILD_DIV=2
run CVO_HIP_BUILD_FIRST=1 CVO_HIP_BUILD_DIV=8
done
echo "#### trace build-first 10k"; CVO_HIP_BUILD_FIRST=1 TAG=ts_bf N=10000 LINES_OUT=8 bash tools/gpu_trace_single.sh
echo "#### trace default 10k"; TAG=ts_df N=10000 LINES_OUT=8 bash tools/gpu_trace_single.sh
