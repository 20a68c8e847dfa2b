#!/bin/bash
export DISTINCT=1 CVO_HIP_ENGINE_DEBUG=1
for v in "2 2" "1000 2" "1000 32"; do
  set -- $v
  echo "== engine_crowd $1 engine_merge_max $2"
  CVO_HIP_ENGINE_CROWD=$1 CVO_HIP_ENGINE_MERGE_MAX=$2 python tools/gpu_batch.py 10000 3 64 2>&1 | grep -E "^B| left |tail:|align_many" | tail -40
done
