#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for n in 200000 50000; do
  python tools/gpu_shard_phases.py $n 8 2>&1 | grep "^shard"
  CVO_HIP_NO_MERGE=1 python tools/gpu_shard_phases.py $n 8 2>&1 | grep "^shard"
done
echo "== kept-list probes: one engine of 22 distinct 10k pairs, MAX_ITER 4 (ell = 0.15) and 21, kernel trace"
for lib in libcvo_hip.so libcvo_hip_nostore.so libcvo_hip_stepexp.so; do
  for mi in 4 21; do
    echo "-- $lib MAX_ITER $mi"
    TAGDIR=gpurun_out/probe_${lib}_$mi; mkdir -p $TAGDIR
    (cd /tmp && export TMPDIR=/tmp && CVO_HIP_ENGINES_FORCE=1 DISTINCT=1 MAX_ITER=$mi CVO_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$TAGDIR -o t -- python $GRAFT_REPO_ROOT/tools/gpu_batch.py 10000 6 22 > $GRAFT_REPO_ROOT/$TAGDIR/log.txt 2>&1)
    tail -1 $TAGDIR/log.txt
    python - <<PY
import csv,glob
f=glob.glob("$TAGDIR/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "kt_process<0, 0>" in r["Name"] or "kt_process<1, 0>" in r["Name"] or "kt_filter" in r["Name"]:
        print("   %-40s calls %6s total %10.1f us avg %8.2f us" % (r["Name"].replace("void cvo_dev::","")[:40], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3))
PY
    rm -rf $TAGDIR/*trace.csv
  done
done
