#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for nb in 64 128 256 1024; do echo "== nblk $nb"; CVO_HIP_PROC_BLOCKS=$nb timeout 120 python tools/gpu_waveload.py 10000 2>&1 | grep "^ell"; done
echo "== 20k nblk 256"; CVO_HIP_PROC_BLOCKS=256 timeout 120 python tools/gpu_waveload.py 20000 2>&1 | grep "^ell"
