#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -m gpu -x -q -k "resident or head_mode or sequence or overflow or stream_of or concurrent or frontend or tum or parity" 2>&1 | grep -E "passed|failed|Error" | head -5
python tools/gpu_r5_run_check.py 3000 10000 2>&1 | grep -v amdgpu | grep "trace    0\|mismatches" | grep "20190402\|mismatches"
for e in "" 1; do echo "NO_FINAL_MIRROR=$e"; if [ -n "$e" ]; then export CVO_HIP_NO_FINAL_MIRROR=1; fi; python tools/gpu_single.py 10000 40 2>&1 | grep "^single"; python tools/gpu_single.py 3000 40 2>&1 | grep "^single"; python tools/gpu_stream.py 2>&1 | grep "cvo prefetch=0 device=True" | tail -1; done
