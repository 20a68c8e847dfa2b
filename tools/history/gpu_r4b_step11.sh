#!/bin/bash
# host-side clocks of the engines for one 64-pair call, full and capped at 4 iterations
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for mi in 4 2000; do
  echo "== max_iter $mi"; MAX_ITER=$mi CVO_HIP_ENGINE_DEBUG=1 PER_STEP=1 DISTINCT=1 CVO_HIP_GRAPH=1 python tools/gpu_batch.py 10000 3 64 2>&1 | tail -24
done 2>&1 | tee gpurun_out/r4b_hostclocks.txt
