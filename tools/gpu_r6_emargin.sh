#!/bin/bash
# list margin of the registrations of a large cvo_hip_align_many call (engines): candidates per member against rebuilds
cd ${GRAFT_REPO_ROOT:-.}
export DISTINCT=1 CVO_HIP_GRAPH=1
for round in 1 2; do
for m in -1 0.12 0.18 0.25 0.35 0.5; do
  echo "== margin $m"; CVO_HIP_LIST_MARGIN=$m python tools/gpu_batch.py 10000 8 64,256 2>&1 | grep "^B" | cut -c1-60
done
done
