#!/bin/bash
# head mode tuning matrix (one registration at a time)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CVO_HIP_GRAPH=1
run() { echo "== $*"; for n in 10000 3000; do env "$@" timeout 120 python tools/gpu_single.py $n 40 cvo 2>&1 | grep "^single"; done; }
run CVO_LIB=libcvo_hip_r2.so
run CVO_HIP_NO_HEAD=1
run CVO_HIP_HEAD_FLUSH=1
run X=1
run CVO_HIP_PACE_LEAD=1
run CVO_HIP_PACE_LEAD=3
run CVO_HIP_BATCH=12
run CVO_HIP_BATCH=16
run CVO_HIP_BATCH=16 CVO_HIP_PACE_LEAD=3
run CVO_HIP_PROC_BLOCKS=512
run CVO_HIP_PROC_BLOCKS=512 CVO_HIP_BATCH=16
run CVO_HIP_PROC_BLOCKS=256
echo "== acvo head"; for n in 10000 3000; do CVO_HIP_HEAD_ACVO=1 timeout 120 python tools/gpu_single.py $n 30 acvo 2>&1 | grep "^single"; done
echo "== acvo head 256"; for n in 10000 3000; do CVO_HIP_HEAD_ACVO=1 CVO_HIP_PROC_BLOCKS=256 timeout 120 python tools/gpu_single.py $n 30 acvo 2>&1 | grep "^single"; done
echo "== acvo classic"; for n in 10000 3000; do timeout 120 python tools/gpu_single.py $n 30 acvo 2>&1 | grep "^single"; done
