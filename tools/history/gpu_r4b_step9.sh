#!/bin/bash
# timeline of a 64-pair call on the current binary: per engine, chain / kernel / gap time by iteration range
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for tag in base merge16; do
  OUT=$ROOTDIR/gpurun_out/r4b_trace_$tag
  mkdir -p $OUT
  if [ $tag = merge16 ]; then export CVO_HIP_MERGE_MAX=16; fi
  DISTINCT=1 CVO_HIP_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $ROOTDIR/tools/gpu_batch.py 10000 3 64 > $OUT/log.txt 2>&1
  tail -1 $OUT/log.txt
  f=$(ls $OUT/*kernel_trace.csv | head -1)
  python $ROOTDIR/tools/trace_engine_phases.py $f 4 | tee $OUT/phases.txt
  python - <<PY
import csv,collections,glob
d=collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    k=r['Kernel_Name'].split('(')[0].replace('cvo_dev::','').replace('void ','')
    d[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
tot=sum(sum(v) for v in d.values())
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    v=sorted(v)
    print("%-28s n %5d  sum %9.1f us (%4.1f %%)  avg %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f"%(k[:28],len(v),sum(v),100*sum(v)/tot,sum(v)/len(v),v[len(v)//2],v[(len(v)*9)//10],v[-1]))
PY
  rm -f $OUT/*kernel_trace.csv $OUT/*agent_info.csv
done
