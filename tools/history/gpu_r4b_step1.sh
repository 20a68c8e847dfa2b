#!/bin/bash
# round 4 (second session), first GPU call: instruction rates, parity of the fused step tail, A/B against the literal tail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 120 tools/microbench/valu_rates > gpurun_out/r4b_valu_rates.txt 2>&1
tail -45 gpurun_out/r4b_valu_rates.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -6
timeout 1200 python tools/gpu_abx_libs.py 3 libcvo_hip.so libcvo_hip_lit.so -- "10000 6 64" "10000 3 256" 2>&1 | tee gpurun_out/r4b_ab_tail.txt
