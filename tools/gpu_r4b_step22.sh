#!/bin/bash
# 8-byte candidate records for clouds of 65 537 ... 262 144 rows
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_paths.py -x -q -k "eight_byte or above_65536 or big or 70k" 2>&1 | tail -3
for r in 1 2; do for v in "" "CVO_HIP_NO_BIG_CAND=1"; do for cfg in "200000 3 cvo" "100000 4 cvo" "70000 5 cvo" "100000 3 acvo"; do echo -n "[$v] "; env $v python tools/gpu_single.py $cfg 2>&1 | grep single; done; done; done | tee gpurun_out/r4b_bigcand.txt
echo "== above 65536 rows soak"; timeout 1800 python tools/gpu_soak_big.py 2>&1 | tail -8
