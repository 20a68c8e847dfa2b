#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for sh in 22 16 12 8 22 18 14; do
  echo "-- CVO_HIP_ENGINE_SHARE=$sh"
  CVO_HIP_ENGINE_SHARE=$sh DISTINCT=1 timeout 300 python tools/gpu_batch.py 10000 8 64 2>&1 | grep "^B " | cut -c1-120
done
for e in 2 4; do for sh in 16 11; do
  echo "-- engines $e share $sh"
  CVO_HIP_ENGINES_FORCE=$e CVO_HIP_ENGINE_SHARE=$sh DISTINCT=1 timeout 300 python tools/gpu_batch.py 10000 8 64 2>&1 | grep "^B " | cut -c1-120
done; done
