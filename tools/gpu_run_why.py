#!/usr/bin/env python3
"""Why resident runs decline (a -DCVO_RUN_WHY build: tools/build_variant.sh why -DCVO_RUN_WHY), per registration, a few
registrations in a row on one context.
usage: CVO_LIB=libcvo_hip_why.so gpu_run_why.py n seed [seed ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), os.environ.get("CVO_LIB", "libcvo_hip_why.so"))
n = int(sys.argv[1])
names = ("done", "stall", "build named", "no record/hint", "done after head", "stall after head", "record not current", "too many", "empty")
for seed in [int(a) for a in sys.argv[2:]]:
    acvo = bool(os.environ.get("ACVO"))
    xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=seed, acvo=acvo)
    c = capi.Context(mode=capi.MODE_ACVO if acvo else capi.MODE_CVO, device=0)
    c.set_fixed(xf, ff); c.set_moving(xm, fm)
    for rep in range(int(os.environ.get("ALIGNS", "4"))):
        c0 = c.run_clocks()
        st = capi.init_state(c.params); n_it, tr = c.align(st, trace_cap=2000 if rep == 0 else 0)
        clk = c.run_clocks(); rs = c.run_stats()
        print("n %d seed %d: %d iterations, runs %d declined %d inside %d; declines: %s" % (n, seed, n_it, rs[0], rs[1], rs[2],
              ", ".join("%s %d" % (nm, v) for nm, v in zip(names, clk) if v)))
        if rep == 0:
            print("   nnz per iteration:", " ".join("%d:%.0fk" % (t["k"], t["nnz"] / 1e3) for t in tr[:n_it:3]))
            print("   step per iteration:", " ".join("%d:%.3f" % (t["k"], t["step"]) for t in tr[:n_it:3]))
    c.close()
