#!/bin/bash
# start-up chains of the launches of one registration: A/B of builds (LIBS = file names in cvo-rgbd_amd/csrc)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
LIBS=${LIBS:-"libcvo_hip_base.so libcvo_hip.so"}
for round in 1 2; do
  for cfg in "3000 60 cvo" "10000 40 cvo" "3000 60 acvo" "10000 40 acvo"; do
    for lib in $LIBS; do
      echo -n "$lib: "; CVO_LIB=$lib python tools/gpu_single.py $cfg 2>&1 | grep single
    done
  done
done
