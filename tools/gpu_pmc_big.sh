#!/bin/bash
# PMC over one large single registration (default 200k x 200k): is PROC_FLOW VALU- or memory-bound?
N=${1:-200000}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/pmc_big
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOTDIR/tools/gpu_batch.py $N 1 1"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('cvo_dev::','').replace('void ','')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,d in sorted(agg.items()):
        if 'rocclr' in k or 'prepare' in k: continue
        print(k, {c: "%.3g"%(sum(v)) for c,v in d.items()}, "launches", len(next(iter(d.values()))))
PY
