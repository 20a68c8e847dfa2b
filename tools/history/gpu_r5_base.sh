#!/bin/bash
# round-5 baseline: one registration at a time, per phase
cd ${GRAFT_REPO_ROOT:-.}
for n in 10000 3000; do python tools/gpu_single_phases.py $n 30 cvo; done
python tools/gpu_single_phases.py 10000 20 acvo
python tools/gpu_single_phases.py 3000 20 acvo
