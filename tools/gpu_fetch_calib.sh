#!/bin/bash
# FETCH_SIZE on a 16-byte gather of known footprint (tools/microbench/fetch_calib.hip): counters in passes of their own
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/fetch_calib
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$ROOTDIR/tools/microbench/fetch_calib
$B > $OUT/plain.txt 2>&1; cat $OUT/plain.txt
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum "TCC_HIT_sum TCC_MISS_sum" TCC_EA0_RDREQ_32B_sum; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$tag -o p -- $B > /dev/null 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  echo "== $c"
  python3 - "$f" <<'PY'
import csv, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in d.items():
    if k.startswith("k_"):
        print("  %-16s " % k + "  ".join("%s per launch (launches 2..6): %.4g" % (c, sum(v[1:]) / max(len(v) - 1, 1)) for c, v in cs.items()))
PY
done
