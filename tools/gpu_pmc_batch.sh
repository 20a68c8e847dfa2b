#!/bin/bash
# PMC passes over a fused batch run (B registrations per launch), per-kernel averages
B=${1:-16}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/pmc_b$B
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
CMD="python $ROOTDIR/tools/gpu_batch.py 10000 3 $B"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('cvo_dev::','').replace('void ','')
        agg[(k,r.get('Grid_Size','') )][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,d in sorted(agg.items()):
        if 'rocclr' in k[0] or 'prepare' in k[0]: continue
        print(k, "p50", {c: round(sorted(v)[len(v)//2],1) for c,v in d.items()}, "p90", {c: round(sorted(v)[(len(v)*9)//10],1) for c,v in d.items()}, "n", len(next(iter(d.values()))))
PY
