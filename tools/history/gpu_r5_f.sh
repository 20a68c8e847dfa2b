#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for lib in libcvo_hip.so libcvo_hip_nostore.so libcvo_hip_stepexp.so; do
  for mi in 4; do
    echo "-- $lib MAX_ITER $mi"
    TAGDIR=gpurun_out/probe_${lib}_$mi; mkdir -p $TAGDIR
    (cd /tmp && export TMPDIR=/tmp && CVO_HIP_ENGINES_FORCE=1 DISTINCT=1 NO_BREAK=1 MAX_ITER=$mi CVO_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$TAGDIR -o t -- python $GRAFT_REPO_ROOT/tools/gpu_batch.py 10000 3 22 > $GRAFT_REPO_ROOT/$TAGDIR/log.txt 2>&1)
    tail -2 $TAGDIR/log.txt
    python - <<PY
import csv,glob
f=glob.glob("$TAGDIR/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
fl=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if "kt_process<0, 0>" in r["Kernel_Name"]]
st=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if "kt_process<1, 0>" in r["Kernel_Name"]]
print("   flow launches (us):", " ".join("%.0f"%x for x in fl[:44]))
print("   step launches (us):", " ".join("%.0f"%x for x in st[:44]))
PY
    rm -rf $TAGDIR/*trace.csv
  done
done
