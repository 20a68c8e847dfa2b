cd ${GRAFT_REPO_ROOT:-.}
# list margins again with the runs' exchanges in two levels (wider lists = more solvers: what an exchange costs by the solver has changed)
for m in -1 0.25 0.5 0.8 1.2; do
  echo "== acvo margin $m"; CVO_HIP_LIST_MARGIN=$m ACVO=1 REPS=30 SEEDS=20190402,1001 python tools/gpu_single_rate.py 3000 6000 10000 14000 2>&1 | grep "^n " | cut -c1-150
done
for m in -1 0.25 0.4 0.6 0.8; do
  echo "== cvo margin $m"; CVO_HIP_LIST_MARGIN=$m REPS=30 SEEDS=20190402,1001 python tools/gpu_single_rate.py 3000 6000 10000 14000 2>&1 | grep "^n " | cut -c1-150
done
