#!/usr/bin/env python3
"""One pair, one registration at a time: with resident runs against without (CVO_HIP_NO_RUN), repeated; where the states differ.
usage: gpu_pair_check.py n seed [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
n, seed = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=seed)
def ctx(no_run):
    if no_run: os.environ["CVO_HIP_NO_RUN"] = "1"
    else: os.environ.pop("CVO_HIP_NO_RUN", None)
    c = capi.Context(mode=capi.MODE_CVO, device=0)
    c.set_fixed(xf, ff); c.set_moving(xm, fm)
    return c
ref_c = ctx(True)
st = capi.init_state(ref_c.params); it_ref, tr_ref = ref_c.align(st, trace_cap=2000); ref = bytes(st)
print("without runs: %d iterations, last exit code %d" % (it_ref, tr_ref[it_ref - 1]["exit_code"] if it_ref <= len(tr_ref) else -1))
c = ctx(False)
for rep in range(reps):
    st = capi.init_state(c.params); it, tr = c.align(st, trace_cap=2000 if rep % 2 == 0 else 0)
    got = bytes(st); rs = c.run_stats()
    diff = [i for i in range(len(ref)) if ref[i] != got[i]]
    print("rep %d: %d iterations, runs %d declined %d inside %d; differing bytes %s" % (rep, it, rs[0], rs[1], rs[2], diff[:12]))
    if diff:
        a = np.frombuffer(ref, np.float32, 62); b = np.frombuffer(got, np.float32, 62)
        for q in sorted(set(i // 4 for i in diff if i < 248)):
            print("   float %d: %r (without) %r (with)" % (q, a[q], b[q]))
