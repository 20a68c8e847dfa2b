#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_paths.py -m gpu -x -q -k "wide_candidate or candidate_list" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config3" 2>&1 | tail -5
export CVO_HIP_GRAPH=1
for n in 200000 100000; do
echo "-- wide off"; CVO_HIP_NO_CAND_WIDE=1 timeout 300 python tools/gpu_single.py $n 3 cvo 2>&1 | grep "^single"
echo "-- wide on";  timeout 300 python tools/gpu_single.py $n 3 cvo 2>&1 | grep "^single"
done
echo "-- acvo 100k wide off"; CVO_HIP_NO_CAND_WIDE=1 timeout 300 python tools/gpu_single.py 100000 3 acvo 2>&1 | grep "^single"
echo "-- acvo 100k wide on";  timeout 300 python tools/gpu_single.py 100000 3 acvo 2>&1 | grep "^single"
