#!/bin/bash
# timeline of the end-to-end stream: where a frame's time goes (GPU busy vs gaps)
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/trace_stream
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- python $ROOTDIR/tools/gpu_stream.py 30 ${1:-acvo} > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
python - <<PY
import csv,glob
rows=[]
for f in glob.glob("$OUT/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('cvo_dev::','').replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:40]))
for f in glob.glob("$OUT/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY'))
rows.sort()
# last 3 frames: a frame starts at k_fe_level0
starts=[i for i,r in enumerate(rows) if r[2]=='k_fe_level0']
a,b=starts[-3],starts[-2]
t0=rows[a][0]
prev=None
for s,e,n in rows[a:b]:
    gap=(s-prev)/1e3 if prev else 0.0
    flag=" <-- gap %.1f"%gap if gap>8 else ""
    print("%8.1f +%6.1f %s%s" % ((s-t0)/1e3,(e-s)/1e3,n,flag))
    prev=e
print("frame period %.1f us" % ((rows[b][0]-t0)/1e3))
PY
