#!/bin/bash
# probe: extra gathers of the same lines in the streaming flow pass (does a round's time follow the number of gathers?)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do for lib in libcvo_hip.so libcvo_hip_xg1.so libcvo_hip_xg2.so libcvo_hip_nt.so; do for mi in 4 21 2000; do
  echo -n "$lib max_iter $mi: "; CVO_LIB=$lib MAX_ITER=$mi DISTINCT=1 CVO_HIP_GRAPH=1 python tools/gpu_batch.py 10000 6 64 2>/dev/null | tail -1 | cut -c1-60
done; done; done 2>&1 | tee gpurun_out/r4b_xg.txt
