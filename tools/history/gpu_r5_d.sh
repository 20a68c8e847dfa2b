#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python tools/gpu_chain.py 120 cvo 2>&1 | grep -v amdgpu | tail -4
python tools/gpu_stream.py 2>&1 | grep -v amdgpu | tail -6
