#!/bin/bash
# kernel trace of a fused / batched run: per-kernel durations at a given batch size
B=${1:-16}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/trace_b$B
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $ROOTDIR/tools/gpu_batch.py ${N:-10000} 5 $B $MODE > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python - <<PY
import csv,collections,glob
f=glob.glob("$OUT/*kernel_trace.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0].replace('cvo_dev::','').replace('void ','')
    d[(k,r['Grid_Size_Z'] if 'Grid_Size_Z' in r else '')].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items()):
    v=sorted(v)
    print("%-28s n %5d  sum %9.1f us  avg %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f"%(str(k),len(v),sum(v),sum(v)/len(v),v[len(v)//10],v[len(v)//2],v[(len(v)*9)//10],v[-1]))
PY
