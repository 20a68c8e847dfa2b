#!/bin/bash
# VALU accounting of a batched run under the variants given as arguments ("ENV=a ENV2=b" strings):
# per kernel, summed over all launches: SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU (quad-cycles), SQ_WAVES, SQ_WAVE_CYCLES.
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/pmc_valu
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for V in "$@"; do
  i=$((i+1))
  env $V DISTINCT=1 CVO_HIP_GRAPH=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/v$i -o p -- python $ROOTDIR/tools/gpu_batch.py ${PMC_N:-10000} 1 ${PMC_B:-64} ${PMC_MODE:-} > $OUT/v$i.log 2>&1
  echo "== variant $i: $V"
  python - <<PY
import csv,collections,glob
f=glob.glob("$OUT/v$i/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0].replace('cvo_dev::','').replace('void ','')[:28]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
tot=collections.defaultdict(float)
for k,d in sorted(agg.items(), key=lambda kv:-kv[1]['SQ_ACTIVE_INST_VALU']):
    if d['SQ_ACTIVE_INST_VALU'] < 1e6: continue
    print("  %-28s" % k, " ".join("%s %.3e" % (c.replace('SQ_',''), d[c]) for c in ('SQ_INSTS_VALU','SQ_ACTIVE_INST_VALU','SQ_WAVES','SQ_WAVE_CYCLES','SQ_INSTS_SALU','SQ_INSTS_LDS')))
    for c in d: tot[c]+=d[c]
print("  %-28s" % "TOTAL", " ".join("%s %.3e" % (c.replace('SQ_',''), tot[c]) for c in ('SQ_INSTS_VALU','SQ_ACTIVE_INST_VALU','SQ_WAVES','SQ_WAVE_CYCLES')))
PY
done
