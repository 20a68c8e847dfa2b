#!/usr/bin/env python3
"""Per-length-scale kernel breakdown of one cvo registration from a rocprofv3
--kernel-trace CSV of tools/gpu_quick.py (second registration = no HIP events)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(r['Kernel_Name'].split('(')[0].replace('cvo_dev::', '').replace('void ', ''),
       int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
ks.sort(key=lambda x: x[1])
preps = [i for i, k in enumerate(ks) if k[0] == 'k_prepare']
seq = [k for k in ks[preps[1]:preps[2]] if not k[0].startswith('__amd')]
fs = [i for i, k in enumerate(seq) if k[0] == 'k_filter']
for rng, name in (((0, 4), 'ell .15'), ((4, 11), 'ell .10'), ((11, 21), 'ell .06'), ((21, 52), 'ell .03')):
    per, parts = [], {}
    for a, b in zip(fs[rng[0]:rng[1]], fs[rng[0] + 1:rng[1] + 1]):
        per.append((seq[b][1] - seq[a][1]) / 1e3)
        for k in seq[a:b]:
            parts.setdefault(k[0], []).append((k[2] - k[1]) / 1e3)
    print(name, 'period %.1f us |' % (sum(per) / len(per)),
          ' '.join('%s %.1f' % (k, sum(v) / len(v)) for k, v in parts.items()))
