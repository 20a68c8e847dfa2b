#!/bin/bash
# how the wall time of a 64-pair call splits into the heavy iterations and the rest: iteration caps
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for mi in 4 11 21 31 41 61 81 121 2000; do
  for B in 64 22; do
    echo -n "max_iter $mi: "; MAX_ITER=$mi DISTINCT=1 CVO_HIP_GRAPH=1 python tools/gpu_batch.py 10000 5 $B | tail -1
  done
done 2>&1 | tee gpurun_out/r4b_itercap.txt
echo "one engine:"; for mi in 21 2000; do echo -n "max_iter $mi: "; MAX_ITER=$mi CVO_HIP_ENGINES_FORCE=1 DISTINCT=1 CVO_HIP_GRAPH=1 python tools/gpu_batch.py 10000 5 22 | tail -1; done 2>&1 | tee -a gpurun_out/r4b_itercap.txt
