#!/bin/bash
# per-launch VALU instruction counts of the list kernels at the four length-scales (variants = env strings)
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/flow_valu
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for V in "$@"; do
  i=$((i+1))
  env $V rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/v$i -o p -- python $ROOTDIR/tools/gpu_flow_valu.py ${PMC_N:-10000} > $OUT/v$i.log 2>&1
  echo "== variant $i: $V"; grep "^ell" $OUT/v$i.log
  python - <<PY
import csv,collections,glob
f=glob.glob("$OUT/v$i/*counter_collection.csv")[0]
rows=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0].replace('cvo_dev::','').replace('void ','')[:24]
    if 'process' not in k and 'filter' not in k: continue
    rows.setdefault((int(r['Dispatch_Id']),k),{})[r['Counter_Name']]=float(r['Counter_Value'])
for (d,k),c in rows.items():
    print("  %4d %-24s VALU %9.0f active %9.0f SALU %9.0f waves %6.0f wave_cycles %10.0f gui %8.0f" % (d,k,c.get('SQ_INSTS_VALU',0),c.get('SQ_ACTIVE_INST_VALU',0),c.get('SQ_INSTS_SALU',0),c.get('SQ_WAVES',0),c.get('SQ_WAVE_CYCLES',0),c.get('GRBM_GUI_ACTIVE',0)))
PY
done
