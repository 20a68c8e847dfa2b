#!/bin/bash
# counters of the flow pass with its rows in LDS against the gathers: the four iterations at ell = 0.15 of 64 pairs
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in libcvo_hip.so libcvo_hip_notile.so; do
  OUT=$ROOTDIR/gpurun_out/r4b_pmc_tiled_$lib; mkdir -p $OUT
  CMD="python $ROOTDIR/tools/gpu_batch.py 10000 2 64"
  export CVO_LIB=$lib MAX_ITER=4 DISTINCT=1 CVO_HIP_GRAPH=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- $CMD > $OUT/t.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o p -- $CMD > $OUT/p1.log 2>&1
  rocprofv3 --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o p -- $CMD > $OUT/p2.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/p3 -o p -- $CMD > $OUT/p3.log 2>&1
  python - <<PY
import csv,glob,collections
out="$OUT"
f=glob.glob(out+"/t/*kernel_trace.csv")[0]
d=[ (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if 'kt_process<0, 0>' in r['Kernel_Name']]
d.sort(); heavy=[x for x in d if x>50]
print("$lib flow launches %d, heavy (>50 us) %d: mean %.1f us, sum %.1f ms" % (len(d), len(heavy), sum(heavy)/max(1,len(heavy)), sum(heavy)/1e3))
for p in ("p1","p2","p3"):
    tot=collections.defaultdict(float)
    for ff in glob.glob(out+"/"+p+"/*counter_collection.csv"):
        for r in csv.DictReader(open(ff)):
            if 'kt_process<0, 0>' in r['Kernel_Name']: tot[r['Counter_Name']]+=float(r['Counter_Value'])
    print("   ", {k: "%.4g" % v for k,v in sorted(tot.items())})
PY
  rm -rf $OUT/t $OUT/p1 $OUT/p2 $OUT/p3
done
