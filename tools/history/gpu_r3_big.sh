#!/bin/bash
# kernel shares of one large registration (BASELINE configs[3] shape on one GPU)
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/big
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CVO_HIP_GRAPH=1
for n in 200000 50000; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$n -o stats -- python $ROOTDIR/tools/gpu_single.py $n 2 ${MODE:-cvo} > $OUT/log$n.txt 2>&1
grep "^single" $OUT/log$n.txt
python - <<PY
import csv,glob
f=glob.glob("$OUT/s$n/*kernel_stats.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if 'cvo_dev' in r['Name']]
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:8]:
    print("  %-60s calls %5s avg %9.1f us  %5.1f %%" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
rm -f $OUT/s$n/*kernel_trace.csv
done
