cd ${GRAFT_REPO_ROOT:-.}
for v in "CVO_HIP_NO_SIDE=1" "CVO_HIP_RUN_BUILD_AT=0.6" "CVO_HIP_RUN_BUILD_AT=0.75" "CVO_HIP_RUN_BUILD_AT=0.9"; do
  echo "== $v cvo"; env $v REPS=30 SEEDS=20190402,1001 python tools/gpu_single_rate.py 3000 6000 10000 2>&1 | grep "^n "
  echo "== $v acvo"; env $v ACVO=1 REPS=30 SEEDS=20190402,1001 python tools/gpu_single_rate.py 3000 6000 10000 2>&1 | grep "^n "
done
