#!/bin/bash
# round-4 A/B (3): ticket tails (post parts in the tails of the list passes' launches) in the engines
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CVO_HIP_GRAPH=1 DISTINCT=1
run() { echo "== $*"; env "$@" timeout 300 python tools/gpu_batch.py 10000 12 64,256 2>&1 | grep "^B"; }
run CVO_HIP_SEGREGATE_MIN=0
run CVO_HIP_SEGREGATE_MIN=0 CVO_HIP_TAILS=1
run CVO_HIP_SEGREGATE_MIN=0
run CVO_HIP_SEGREGATE_MIN=0 CVO_HIP_TAILS=1
run CVO_HIP_TAILS=1 CVO_HIP_COHORT=11 CVO_HIP_HEAVY_ENGINES=2 CVO_HIP_ENGINES_FORCE=4
echo "== acvo"; 
for t in 0 1; do echo "-- tails $t"; if [ $t = 1 ]; then export CVO_HIP_TAILS=1; else unset CVO_HIP_TAILS; fi; CVO_HIP_SEGREGATE_MIN=0 timeout 300 python tools/gpu_batch.py 10000 8 64 acvo 2>&1 | grep "^B"; done
echo "== 3k"; 
for t in 0 1; do echo "-- tails $t"; if [ $t = 1 ]; then export CVO_HIP_TAILS=1; else unset CVO_HIP_TAILS; fi; CVO_HIP_SEGREGATE_MIN=0 timeout 300 python tools/gpu_batch.py 3000 12 64 2>&1 | grep "^B"; done
echo "== 20k x 8"; 
for t in 0 1; do echo "-- tails $t"; if [ $t = 1 ]; then export CVO_HIP_TAILS=1; else unset CVO_HIP_TAILS; fi; CVO_HIP_SEGREGATE_MIN=0 timeout 300 python tools/gpu_batch.py 20000 8 8 2>&1 | grep "^B"; done
export CVO_HIP_TAILS=1
echo "== tests with tails"
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_paths.py -x -q -k "align_many or headline or fused or refills or mixed_bag or engine_profiling or config4" 2>&1 | tail -5
