#!/bin/bash
# timeline of one short align() (max_iter 2): where the fixed cost of a registration goes
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/trace_ovh
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/ovh.py <<PY
import os, sys, time
sys.path.insert(0, "$ROOTDIR")
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
xf, ff, xm, fm = pkg.data.synthetic_pair(10000, 10000, seed=pkg.data.SEED_CFG2)
prm = capi.default_params(capi.MODE_CVO); prm.max_iter = 2
c = capi.Context(mode=capi.MODE_CVO, device=0, params=prm)
c.set_fixed(xf, ff); c.set_moving(xm, fm)
for _ in range(6):
    st = capi.init_state(c.params); c.align(st, trace_cap=0)
    time.sleep(0.002)
c.close()
PY
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- python /tmp/ovh.py > $OUT/log.txt 2>&1
python - <<PY
import csv,glob
rows=[]
for f in glob.glob("$OUT/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('cvo_dev::','').split('(')[0]))
for f in glob.glob("$OUT/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY '+r.get('Direction','')))
rows.sort()
# the last align: find the last k_prepare
ip=[i for i,r in enumerate(rows) if r[2]=='k_prepare'][-1]
t0=rows[ip-2][0] if ip>=2 else rows[ip][0]
for s,e,n in rows[max(0,ip-3):ip+40]:
    print("%8.1f us  +%6.1f  %s" % ((s-t0)/1e3, (e-s)/1e3, n))
PY
