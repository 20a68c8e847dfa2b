#!/bin/bash
# two-level exchanges from g solvers on (-DCVO_RUN_HIER_FROM): 9 / 33 (built) / 65 / 129 / never, two interleaved rounds
cd ${GRAFT_REPO_ROOT:-.}
for round in 1 2; do
for lib in libcvo_hip.so libcvo_hip_pl2.so libcvo_hip_pl1.so; do
  echo "== $lib"
  CVO_LIB=$lib REPS=40 SEEDS=20190402,1001 python tools/gpu_single_rate.py 3000 6000 10000 14000 2>&1 | grep "^n " | cut -c1-70
  CVO_LIB=$lib ACVO=1 REPS=40 SEEDS=20190402,1001 python tools/gpu_single_rate.py 3000 6000 10000 14000 2>&1 | grep "^n " | cut -c1-70
done
done
