#!/bin/bash
# one registration at a time, two (or more) builds alternating: LIBS = file names in cvo-rgbd_amd/csrc
cd ${GRAFT_REPO_ROOT:-$(pwd)}
LIBS=${LIBS:-"libcvo_hip_base.so libcvo_hip.so"}
python -m pytest tests/test_gpu_paths.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2
for round in 1 2 3; do
  for cfg in "3000 60 cvo" "6000 40 cvo" "10000 40 cvo" "14000 30 cvo" "3000 60 acvo" "10000 40 acvo"; do
    for lib in $LIBS; do
      echo -n "$lib: "; CVO_LIB=$lib python tools/gpu_single.py $cfg 2>&1 | grep single
    done
  done
done
