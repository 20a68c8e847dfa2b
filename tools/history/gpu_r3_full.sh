cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
SOAK_SEED=2024 timeout 900 python tools/gpu_soak.py 800 4000 2>&1 | tail -1
SOAK_DEGENERATE=1 SOAK_SEED=11 timeout 900 python tools/gpu_soak.py 400 3000 2>&1 | tail -1
SOAK_SEED=7 timeout 900 python tools/gpu_soak.py 150 12000 2>&1 | tail -1
CVO_HIP_LIST_INIT=65536 SOAK_SEED=3 timeout 900 python tools/gpu_soak.py 200 3000 2>&1 | tail -1
