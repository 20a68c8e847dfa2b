#!/bin/bash
# round-3 working check: GPU suite, then single-stream rates (new default / no head / round-2 library)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
export CVO_HIP_GRAPH=1
for n in 10000 3000; do
 for m in cvo acvo; do
  [ -f cvo-rgbd_amd/csrc/libcvo_hip_r2.so ] && { echo "-- r2 lib"; CVO_LIB=libcvo_hip_r2.so timeout 120 python tools/gpu_single.py $n 30 $m 2>&1 | grep "^single"; }
  echo "-- classic";  CVO_HIP_NO_HEAD=1 timeout 120 python tools/gpu_single.py $n 30 $m 2>&1 | grep "^single"
  echo "-- head";     timeout 120 python tools/gpu_single.py $n 30 $m 2>&1 | grep "^single"
 done
done
