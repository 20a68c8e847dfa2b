#!/usr/bin/env python3
"""Interleaved A/B on the GPU box: each variant (a set of environment variables) runs tools/gpu_batch.py
`rounds` times in turn; prints the median registrations/s per variant and workload.
usage: gpu_abx.py rounds "ENV1=a ENV2=b" "ENV1=c" ... -- "10000 10 1" "10000 5 32" ..."""
import os, re, subprocess, sys, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds = int(args[0]); sep = args.index("--")
variants, loads = args[1:sep], args[sep + 1:]
res = {}
for r in range(rounds):
    for v in variants:
        env = dict(os.environ, CVO_HIP_GRAPH="1")
        for kv in v.split():
            if "=" in kv:
                k, x = kv.split("=", 1); env[k] = x
        for l in loads:
            env.setdefault("DISTINCT", "1"); out = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_batch.py")] + l.split(), env=env, capture_output=True, text=True).stdout
            for m in re.finditer(r"B +(\d+): ([0-9.]+) registrations/s", out):
                res.setdefault((v, l, m.group(1)), []).append(float(m.group(2)))
for (v, l, b), xs in res.items():
    print("%-40s | %-22s B %3s | median %8.1f  (%s)" % (v, l, b, statistics.median(xs), " ".join("%.0f" % x for x in xs)))
