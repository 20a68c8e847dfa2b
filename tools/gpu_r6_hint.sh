#!/bin/bash
# a run's launch sized by the expected record ("launch_by_hint"), off / on, interleaved; then the small calls with the growing share on top
cd ${GRAFT_REPO_ROOT:-.}
for round in 1 2; do
for v in off on; do
  [ $v = off ] && export CVO_HIP_NO_LAUNCH_BY_HINT=1 || unset CVO_HIP_NO_LAUNCH_BY_HINT
  echo "== $v"
  REPS=40 SEEDS=20190402,1001 python tools/gpu_single_rate.py 3000 6000 10000 14000 2>&1 | grep "^n " | cut -c1-70
  ACVO=1 REPS=40 SEEDS=20190402,1001 python tools/gpu_single_rate.py 3000 6000 10000 14000 2>&1 | grep "^n " | cut -c1-70
done
done
unset CVO_HIP_NO_LAUNCH_BY_HINT
export DISTINCT=1
for round in 1 2; do
for v in "1 1" "0 1" "0 0"; do
  set -- $v
  [ $1 = 1 ] && export CVO_HIP_NO_LAUNCH_BY_HINT=1 || unset CVO_HIP_NO_LAUNCH_BY_HINT
  [ $2 = 1 ] && export CVO_HIP_NO_SHARE_GROW=1 || unset CVO_HIP_NO_SHARE_GROW
  echo "== small calls: no_hint $1 no_grow $2"
  python tools/gpu_batch.py 3000 10 2,4,8 2>&1 | grep "^B" | cut -c1-40
  python tools/gpu_batch.py 10000 8 2,4,8 2>&1 | grep "^B" | cut -c1-40
done
done
