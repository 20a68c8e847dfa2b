#!/usr/bin/env python3
"""Where the time of an align_many call goes, from a rocprofv3 --kernel-trace CSV of tools/gpu_batch.py:
the LAST call of the trace, per engine stream, iteration = from one flow launch (kt_process<0, 0>) to the
next; per range of iterations the mean chain time, the kernel time inside it and the gaps between launches.
usage: trace_engine_phases.py <kernel_trace.csv> <calls in the trace>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ev = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("cvo_dev::", "").replace("void ", "")
    q = r.get("Queue_Id", r.get("Stream_Id", "0"))
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, q))
ev.sort()
byq = collections.defaultdict(list)
for e in ev:
    if e[2].startswith("kt_"):
        byq[e[3]].append(e)
buckets = [(0, 3), (3, 10), (10, 20), (20, 40), (40, 60), (60, 80), (80, 100), (100, 130), (130, 200)]
print("queue | iterations | n | chain us (mean) | kernel us in it | gaps us | launches per iteration")
total_wall = {}
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    flows = [i for i, e in enumerate(lst) if e[2].startswith("kt_process<0, 0>")]
    if len(flows) < 50:
        continue
    # calls: a gap of > 300 us before a flow launch starts a new call
    starts = [0]
    for a, b in zip(flows[:-1], flows[1:]):
        if lst[b][0] - lst[b - 1][1] > 300e3:
            starts.append(flows.index(b))
    first = starts[-1]
    fl = flows[first:]
    print("queue %s: %d flow launches in the last call, call span %.2f ms" % (q, len(fl), (lst[-1][1] - lst[fl[0]][0]) / 1e6))
    for lo, hi in buckets:
        ch, kt, gp, nl = [], [], [], []
        for k in range(lo, min(hi, len(fl) - 1)):
            a, b = fl[k], fl[k + 1]
            seg = lst[a:b]
            # (the filter launch of iteration k+1 precedes its flow launch: it belongs to the next chain; close enough)
            ch.append((lst[b][0] - lst[a][0]) / 1e3)
            kt.append(sum(e[1] - e[0] for e in seg) / 1e3)
            gp.append(ch[-1] - kt[-1])
            nl.append(len(seg))
        if ch:
            print("   it %3d-%3d | n %3d | chain %8.1f | kernels %8.1f | gaps %7.1f | %.1f launches | sum %.2f ms" % (
                lo, hi, len(ch), sum(ch) / len(ch), sum(kt) / len(kt), sum(gp) / len(gp), sum(nl) / len(nl), sum(ch) / 1e3))
