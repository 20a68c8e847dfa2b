cd ${GRAFT_REPO_ROOT:-.}
for lib in libcvo_hip.so libcvo_hip_pl3.so libcvo_hip_pl4.so; do
echo "== $lib"; CVO_LIB=$lib ACVO=1 REPS=40 SEEDS=20190402,1001 python tools/gpu_single_rate.py 3000 6000 10000 14000 2>&1 | grep "^n "
done
