#!/bin/bash
# round 5: soak of the resident runs (kt_run) against the oracle, new seeds; the run statistics say they were used
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "== random 600"; SOAK_SEED=60601 timeout 2400 python tools/gpu_soak.py 600 4000 2>&1 | tail -2
echo "== degenerate 320"; SOAK_DEGENERATE=1 SOAK_SEED=60602 timeout 1800 python tools/gpu_soak.py 320 3000 2>&1 | tail -2
echo "== larger clouds 100 x 12000"; SOAK_SEED=60603 timeout 2400 python tools/gpu_soak.py 100 12000 2>&1 | tail -2
echo "== graphs on, 300"; CVO_HIP_GRAPH=1 SOAK_SEED=60604 timeout 1800 python tools/gpu_soak.py 300 3500 2>&1 | tail -2
