cd ${GRAFT_REPO_ROOT:-.}
echo "== cvo"; REPS=40 python tools/gpu_single_rate.py 3000 6000 10000 2>&1 | grep "^n "
echo "== acvo"; ACVO=1 REPS=40 python tools/gpu_single_rate.py 3000 6000 10000 2>&1 | grep "^n "
echo "== clocks cvo"; CVO_LIB=libcvo_hip_clk.so python tools/gpu_run_clocks.py 3000 10000 2>&1 | tail -6
echo "== clocks acvo"; ACVO=1 CVO_LIB=libcvo_hip_clk.so python tools/gpu_run_clocks.py 3000 2>&1 | tail -3
