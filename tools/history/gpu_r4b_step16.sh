#!/bin/bash
# clocks inside the head-mode launches of one registration at a time
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "10000 20 cvo" "3000 20 cvo"; do CVO_HIP_POST_DEBUG=1 python tools/gpu_single.py $cfg 2>&1 | grep -v amdgpu.ids | tail -6; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr1 -o t -- python $GRAFT_REPO_ROOT/tools/gpu_single.py 10000 20 cvo > /tmp/tr1.log 2>&1
python - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/tr1/*kernel_trace.csv')[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0].replace('cvo_dev::','')) for r in csv.DictReader(open(f))]
rows.sort()
# steady state: pairs hflow -> step_twist -> hflow
gaps=collections.defaultdict(list); durs=collections.defaultdict(list)
for (s0,e0,n0),(s1,e1,n1) in zip(rows[:-1],rows[1:]):
    if n0.startswith('kt_') and n1.startswith('kt_'):
        gaps[(n0[:16],n1[:16])].append((s1-e0)/1e3); durs[n0[:16]].append((e0-s0)/1e3)
for k,v in gaps.items():
    v.sort(); print('gap',k,'n',len(v),'p50 %.2f us p90 %.2f'%(v[len(v)//2],v[len(v)*9//10]))
for k,v in durs.items():
    v.sort(); print('dur',k,'n',len(v),'p10 %.2f p50 %.2f p90 %.2f'%(v[len(v)//10],v[len(v)//2],v[len(v)*9//10]))
PY
