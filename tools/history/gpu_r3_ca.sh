#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_paths.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "head_mode or parity or config1 or overflow or reuse or candidate" 2>&1 | tail -4
export CVO_HIP_GRAPH=1
run() { echo "== $*"; for n in 10000 3000 6000; do for m in cvo acvo; do env "$@" timeout 120 python tools/gpu_single.py $n 40 $m 2>&1 | grep "^single"; done; done; }
for r in 1 2; do
run CVO_HIP_NO_CAND_ASYNC=1
run X=1
done
