#!/bin/bash
# 8-byte kept entries for clouds of 65 537 ... 262 144 rows (default) against 8 + 4 bytes (CVO_HIP_NO_PACK_WIDE=1)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python -m pytest tests/test_gpu_paths.py -m gpu -x -q -k "eight_byte or wide_candidate" 2>&1 | grep -E "passed|failed|error|Error" | tail -3
python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config3" 2>&1 | grep -E "passed|failed|error" | tail -2
for r in 1 2; do
for cfg in "200000 3 cvo" "100000 5 cvo" "100000 5 acvo" "70000 6 cvo"; do
  echo -n "8+4: "; CVO_HIP_NO_PACK_WIDE=1 python tools/gpu_single.py $cfg 2>&1 | grep single
  echo -n "8  : "; python tools/gpu_single.py $cfg 2>&1 | grep single
done
done
