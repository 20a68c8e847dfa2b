#!/bin/bash
# one registration at a time under the kernel trace: the launches of the last align() with start, duration and the gap in front
# usage: gpu_timeline.sh [n]
cd ${GRAFT_REPO_ROOT:-.}
N=${1:-10000}
python tools/gpu_single_phases.py $N 30 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm"
D=gpurun_out/timeline_$N; mkdir -p $D
cat > /tmp/one.py <<PY
import os, sys
sys.path.insert(0, "$PWD")
import torch, __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
xf, ff, xm, fm = pkg.data.synthetic_pair($N, $N, seed=pkg.data.SEED_CFG2, acvo=bool(os.environ.get("ACVO")))
c = capi.Context(mode=capi.MODE_ACVO if os.environ.get("ACVO") else capi.MODE_CVO, device=0)
c.set_fixed(xf, ff); c.set_moving(xm, fm)
for _ in range(6):
    st = capi.init_state(c.params); n_it, _ = c.align(st, trace_cap=0)
    torch.cuda.synchronize()
print("iterations", n_it)
PY
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$D -o t -- python /tmp/one.py > $GRAFT_REPO_ROOT/$D/log.txt 2>&1)
python - <<PY
import csv, glob, re
f = glob.glob("$D/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last align: launches after the last k_prepare
idx = max(i for i, r in enumerate(rows) if "k_prepare" in r["Kernel_Name"])
rows = rows[idx:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
def short(nm):
    nm = re.sub(r"\(.*", "", nm); nm = nm.replace("cvo_dev::", "").replace("void ", "")
    return nm[:44]
tot = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  %7.1f us  gap %6.1f  %s  grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short(r["Kernel_Name"]), r.get("Grid_Size", "")))
    k = short(r["Kernel_Name"]); a = tot.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3; a[2] += max(0, (s - prev_end)) / 1e3
    prev_end = max(prev_end, e)
print("total %.1f us" % ((prev_end - t0) / 1e3))
for k, a in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("  %-44s launches %3d  busy %7.1f us  gaps in front %7.1f us" % (k, a[0], a[1], a[2]))
PY
rm -f $D/*trace.csv
