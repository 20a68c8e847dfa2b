#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python tools/gpu_abx.py 3 "CVO_LIB=libcvo_hip_base.so" "CVO_LIB=libcvo_hip_nt.so" -- "10000 6 64" "10000 3 256" "20000 4 8"
