#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python tools/gpu_r5_run_check.py 3000 6000 10000 2>&1 | tail -8
python tools/gpu_single_rate.py 3000 6000 10000 14000 2>&1 | grep "^n "
python -m pytest tests -m gpu -x -q -k "resident or on_its_own or parity or config" 2>&1 | tail -3
