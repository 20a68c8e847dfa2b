#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -m gpu -x -q -k "align_many or overflow or reuse or fused or mixed_bag or refills or config or soak or parity or ranks or matlab" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
for b in 64 256; do
  echo "-- FUSED_FILTER_BLOCKS $b: one engine of 22 distinct pairs, kt_filter launches (us)"
  TAGDIR=gpurun_out/filt_$b; mkdir -p $TAGDIR
  (cd /tmp && export TMPDIR=/tmp && CVO_HIP_FUSED_FILTER_BLOCKS=$b CVO_HIP_ENGINES_FORCE=1 DISTINCT=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$TAGDIR -o t -- python $GRAFT_REPO_ROOT/tools/gpu_batch.py 10000 3 22 > $GRAFT_REPO_ROOT/$TAGDIR/log.txt 2>&1)
  grep "^B " $TAGDIR/log.txt
  python - <<PY
import csv,glob
f=glob.glob("$TAGDIR/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
fl=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if "kt_filter" in r["Kernel_Name"]]
big=[x for x in fl if x>30]; small=[x for x in fl if x<=30]
print("   launches %d: builds (> 30 us) %d avg %.1f max %.1f; the others %d avg %.2f; total %.0f us" % (len(fl), len(big), sum(big)/max(len(big),1), max(big or [0]), len(small), sum(small)/max(len(small),1), sum(fl)))
print("   first 40:", " ".join("%.0f"%x for x in fl[:40]))
PY
  rm -rf $TAGDIR/*trace.csv
done
