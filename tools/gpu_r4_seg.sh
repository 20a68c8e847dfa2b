#!/bin/bash
# round-4 A/B: phase-segregated engines (heavy cohort engine + light engines) vs the round-3 engines
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CVO_HIP_GRAPH=1 DISTINCT=1
run() { echo "== $*"; env "$@" timeout 300 python tools/gpu_batch.py 10000 12 64,256 2>&1 | grep "^B"; }
run CVO_HIP_SEGREGATE_MIN=0
run CVO_HIP_COHORT=8
run CVO_HIP_COHORT=4
run CVO_HIP_COHORT=12
run CVO_HIP_COHORT=16
run CVO_HIP_COHORT=8 CVO_HIP_HEAVY_ENGINES=2 CVO_HIP_ENGINES_FORCE=4
run CVO_HIP_COHORT=8 CVO_HIP_ENGINES_FORCE=4
run CVO_HIP_COHORT=8 CVO_HIP_HEAVY_BATCHES=1
run CVO_HIP_COHORT=8 CVO_HIP_HEAVY_BATCHES=3
echo "== debug"; CVO_HIP_ENGINE_DEBUG=1 CVO_HIP_COHORT=8 timeout 300 python tools/gpu_batch.py 10000 3 64 2>&1 | tail -12
