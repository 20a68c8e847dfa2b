#!/usr/bin/env python3
"""One 200k x 200k registration stopped after MAX_ITER iterations (default 1: the expansion pass over the fresh list at ell = 0.15,
the same inputs in every build): what rocprofv3's kernel trace is run on for the kept-list probes (profiles/r06_ab.txt 6).
usage: CVO_LIB=... MAX_ITER=1 gpu_r6_big_probe.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
if os.environ.get("CVO_LIB"):
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), os.environ["CVO_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=pkg.data.SEED_CFG4)
p = capi.default_params(capi.MODE_CVO); p.max_iter = int(os.environ.get("MAX_ITER", "1"))
c = capi.Context(mode=capi.MODE_CVO, device=0, params=p)
c.set_fixed(xf, ff); c.set_moving(xm, fm)
for rep in range(3):
    st = capi.init_state(c.params); torch.cuda.synchronize(); t = time.perf_counter()
    it, _ = c.align(st, trace_cap=0); torch.cuda.synchronize()
    print("rep %d: %d iterations, %.2f ms" % (rep, it, (time.perf_counter() - t) * 1e3))
c.close()
