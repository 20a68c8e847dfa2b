#!/bin/bash
# kept-list probes on ONE 200k x 200k registration (profiles/r06_ab.txt 6): per-kernel durations of iterations 0 .. MAX_ITER-1 in three builds
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
for lib in libcvo_hip.so libcvo_hip_nokept.so libcvo_hip_stepexp.so; do
  for mi in 1 4; do
    D=$R/gpurun_out/bigprobe_${lib%.so}_$mi; mkdir -p $D
    (cd /tmp && export TMPDIR=/tmp && CVO_LIB=$lib MAX_ITER=$mi rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $R/tools/gpu_r6_big_probe.py > $D/log.txt 2>&1)
    echo "== $lib max_iter $mi: $(grep '^rep 2' $D/log.txt)"
    python - <<PY
import csv, glob, re, collections
f = glob.glob("$D/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "k_prepare" in r["Kernel_Name"])
tot = collections.OrderedDict()
for r in rows[idx:]:
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("cvo_dev::", "").replace("void ", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d > 20.0: tot.setdefault(k, []).append(d)
for k, v in tot.items(): print("   %-40s %s us" % (k, " ".join("%.0f" % x for x in v)))
PY
    rm -f $D/*trace.csv
  done
done
