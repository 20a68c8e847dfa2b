#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV: per queue busy time and gaps, the union
busy time of the device, and per (kernel, grid z) duration statistics.
usage: trace_timeline.py <kernel_trace.csv> [skip_first_ms | -keep_last_ms]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0
ev = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("cvo_dev::", "").replace("void ", "")
    if name.startswith("__amd"):
        continue
    q = r.get("Queue_Id", r.get("Stream_Id", "0"))
    z = r.get("Grid_Size_Z", r.get("Grid_Size", ""))
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, q, z))
ev.sort()
t0 = ev[0][0] + skip if skip >= 0 else ev[-1][1] + skip   # negative: keep the last |skip| ms
ev = [e for e in ev if e[0] >= t0]
span = (ev[-1][1] - ev[0][0]) / 1e3
print("kernels %d, span %.1f us" % (len(ev), span))
# union busy
busy, cur_s, cur_e = 0.0, ev[0][0], ev[0][1]
for s, e, *_ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("device busy (union of kernel intervals): %.1f us = %.1f %% of the span" % (busy / 1e3, busy / 10.0 / span))
byq = collections.defaultdict(list)
for e in ev:
    byq[e[3]].append(e)
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    if len(lst) < 20:
        continue
    b = sum(e[1] - e[0] for e in lst) / 1e3
    gaps = [(lst[i + 1][0] - lst[i][1]) / 1e3 for i in range(len(lst) - 1)]
    big = sorted([g for g in gaps if g > 20.0], reverse=True)
    print("queue %s: %d kernels, busy %.1f us, span %.1f us, gaps: median %.2f us, sum of gaps > 20 us: %.1f us (%d), largest %s"
          % (q, len(lst), b, (lst[-1][1] - lst[0][0]) / 1e3, sorted(gaps)[len(gaps) // 2] if gaps else 0.0,
             sum(big), len(big), ["%.0f" % g for g in big[:6]]))
d = collections.defaultdict(list)
for s, e, name, q, z in ev:
    d[(name, z)].append((e - s) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print("%-22s z %-3s n %6d  sum %10.1f us (%4.1f %%)  avg %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f"
          % (k[0], k[1], len(v), sum(v), 100.0 * sum(v) / tot, sum(v) / len(v), v[len(v) // 10], v[len(v) // 2],
             v[(len(v) * 9) // 10], v[-1]))
