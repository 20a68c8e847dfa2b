#!/bin/bash
# XCD-aware slots in the engines' list passes (kt_process: all blocks of a slot on one XCD) against the plain mapping (-DCVO_NO_XCD_SLOTS), interleaved
cd ${GRAFT_REPO_ROOT:-.}
export DISTINCT=1 CVO_HIP_GRAPH=1
for round in 1 2 3; do
  for lib in libcvo_hip_noxcd.so libcvo_hip.so; do
    echo "== $lib"; CVO_LIB=$lib python tools/gpu_batch.py 10000 8 64,256 2>&1 | grep "^B" | cut -c1-150
  done
done
for lib in libcvo_hip_noxcd.so libcvo_hip.so; do
  echo "== acvo $lib"; CVO_LIB=$lib python tools/gpu_batch.py 10000 6 64 acvo 2>&1 | grep "^B" | cut -c1-150
  echo "== 3k $lib"; CVO_LIB=$lib python tools/gpu_batch.py 3000 8 64 2>&1 | grep "^B" | cut -c1-150
  echo "== 20k $lib"; CVO_LIB=$lib python tools/gpu_batch.py 20000 4 8 2>&1 | grep "^B" | cut -c1-150
done
