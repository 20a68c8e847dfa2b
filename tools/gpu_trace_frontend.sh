#!/bin/bash
# kernel trace of the front end: per-kernel durations over 100 frames
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/trace_fe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $ROOTDIR/tools/gpu_frontend.py 100 ${1:-1.0} > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python - <<PY
import csv,collections,glob
f=glob.glob("$OUT/*kernel_trace.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','').split('(')[0]
    d[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
tot=0
for k,v in sorted(d.items()):
    v=sorted(v); tot+=sum(v)
    print("%-28s n %5d  sum %9.1f us  avg %7.2f  p50 %7.2f  max %7.2f"%(k,len(v),sum(v),sum(v)/len(v),v[len(v)//2],v[-1]))
print("kernel time per frame: %.1f us" % (tot/108))
PY
