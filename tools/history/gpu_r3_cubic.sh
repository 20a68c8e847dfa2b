#!/bin/bash
# the sectioning short cut (se3_math.hpp section_shortcut): parity, soak, one registration at a time
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python -m pytest tests/test_gpu_paths.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
SOAK_SEED=777 timeout 900 python tools/gpu_soak.py 600 4000 2>&1 | tail -1
for r in 1 2; do
for cfg in "3000 60 cvo" "6000 40 cvo" "10000 40 cvo" "3000 60 acvo" "10000 40 acvo"; do python tools/gpu_single.py $cfg 2>&1 | grep single; done
done
DISTINCT=1 CVO_HIP_GRAPH=1 python tools/gpu_batch.py 10000 6 64 2>&1 | grep "registrations/s" | tail -1
CVO_HIP_POST_DEBUG=1 python tools/gpu_single.py 10000 20 cvo 2>&1 | grep -i "post\|head\|cubic\|clk" | tail -5
