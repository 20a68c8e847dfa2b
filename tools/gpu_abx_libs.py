#!/usr/bin/env python3
"""As gpu_abx.py, the variants being BUILDS: file names in cvo-rgbd_amd/csrc (CVO_LIB of tools/gpu_batch.py).
usage: gpu_abx_libs.py rounds libA.so libB.so ... -- "10000 6 64" ..."""
import os, re, subprocess, sys, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds = int(args[0]); sep = args.index("--")
libs, loads = args[1:sep], args[sep + 1:]
res = {}
for r in range(rounds):
    for v in libs:
        env = dict(os.environ, CVO_HIP_GRAPH="1", CVO_LIB=v)
        env.setdefault("DISTINCT", "1")
        for l in loads:
            out = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_batch.py")] + l.split(), env=env, capture_output=True, text=True).stdout
            for m in re.finditer(r"B +(\d+): ([0-9.]+) registrations/s", out):
                res.setdefault((v, l, m.group(1)), []).append(float(m.group(2)))
for (v, l, b), xs in res.items():
    print("%-28s | %-22s B %3s | median %8.1f  (%s)" % (v, l, b, statistics.median(xs), " ".join("%.0f" % x for x in xs)))
