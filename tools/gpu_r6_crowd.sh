#!/bin/bash
# engines of a 64-registration call: synchronous lists + five launches per iteration (crowded, the default) against asynchronous
# builds (engine_crowd large) with and without the twist in the step launch (engine_merge_max)
export DISTINCT=1
for round in 1 2; do
for v in "2 2" "1000 2" "1000 32" "2 32"; do
  set -- $v
  echo "== engine_crowd $1 engine_merge_max $2"
  CVO_HIP_ENGINE_CROWD=$1 CVO_HIP_ENGINE_MERGE_MAX=$2 python tools/gpu_batch.py 10000 8 64,256 2>&1 | grep "^B"
done
done
