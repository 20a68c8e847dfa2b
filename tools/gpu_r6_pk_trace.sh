#!/bin/bash
# per-launch durations of the engines' step pass, one member per trip (built) against two with packed float32 (libcvo_hip_two.so: -DCVO_STEP_TWO)
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/r6_pk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DISTINCT=1 CVO_HIP_GRAPH=1
for lib in libcvo_hip.so libcvo_hip_two.so; do
  CVO_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$lib -o s -- python $ROOTDIR/tools/gpu_batch.py 10000 3 64 > $OUT/$lib.log 2>&1
  echo "== $lib"; grep "^B" $OUT/$lib.log | cut -c1-80
  python - "$OUT/$lib" <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("cvo_dev::", "").replace("void ", "")[:34]
    if n.startswith("kt_"): print("   %-34s calls %6s  total %9.1f us  avg %8.2f us  %5.1f %%" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
P
done
find $OUT -name "*.csv" -size +8M -delete
