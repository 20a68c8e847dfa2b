#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CVO_HIP_GRAPH=1
for N in 2000 3000 4500 6000 8000 10000 14000; do for M in cvo acvo; do timeout 120 python tools/gpu_single.py $N 30 $M 2>&1 | grep "^single"; done; done
timeout 900 python -m pytest tests/test_gpu_paths.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_frontend.py -m gpu -x -q 2>&1 | grep "passed\|failed"
