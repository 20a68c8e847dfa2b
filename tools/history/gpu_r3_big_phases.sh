#!/bin/bash
# one 200k x 200k registration: kernel time by kernel and by range of iterations (rocprofv3 kernel trace)
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/bigph
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CVO_HIP_GRAPH=1
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $ROOTDIR/tools/gpu_single.py 200000 1 cvo > $OUT/log.txt 2>&1
grep "^single" $OUT/log.txt
python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if 'cvo_dev::k' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the LAST registration: count flow launches backwards
name=lambda r:r['Kernel_Name'].split('(')[0].replace('cvo_dev::','').replace('void ','')
flows=[i for i,r in enumerate(rows) if name(r).startswith(('kt_process<0','kt_flow','kt_hflow'))]
# registrations are separated by k_prepare
preps=[i for i,r in enumerate(rows) if name(r).startswith('k_prepare')]
start=preps[-1]
seg=rows[start:]
it=-1; agg=collections.defaultdict(lambda: collections.defaultdict(float))
buckets=[(0,3),(3,10),(10,20),(20,40),(40,200)]
for r in seg:
    n=name(r)
    if n.startswith(('kt_process<0','kt_flow','kt_hflow')): it+=1
    k=max(it,0)
    b=[x for x in buckets if x[0]<=k<x[1]][0]
    agg[b][n]+= (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
for b in buckets:
    tot=sum(agg[b].values())
    print("iterations %3d-%3d: %9.1f us  " % (b[0],b[1],tot) + "  ".join("%s %.0f" % (k,v) for k,v in sorted(agg[b].items(), key=lambda kv:-kv[1])[:6]))
PY
