#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for n in 3000 6000; do for b in 256 512 1024; do echo "n $n PROC_BLOCKS $b"; CVO_HIP_PROC_BLOCKS=$b python tools/gpu_single_phases.py $n 20 cvo 2>&1 | grep -v amdgpu.ids; done; done
