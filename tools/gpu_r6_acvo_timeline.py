import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libcvo_hip_why.so")
names = ("done", "stall", "build named", "no record/hint", "done after head", "stall after head", "record not current", "too many", "empty")
n = 10000
xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=pkg.data.SEED_CFG2, acvo=True)
c = capi.Context(mode=capi.MODE_ACVO, device=0)
c.set_fixed(xf, ff); c.set_moving(xm, fm)
for rep in range(3):
    st = capi.init_state(c.params); n_it, tr = c.align(st, trace_cap=2000)
    clk = c.run_clocks(); rs = c.run_stats()
    print("%d iterations, runs %d declined %d inside %d; %s" % (n_it, rs[0], rs[1], rs[2], ", ".join("%s %d" % (nm, v) for nm, v in zip(names, clk) if v)))
print(sorted(tr[0].keys()))
for t in tr[:20]:
    print(t["k"], "ell %.4f nnz %d xx %d yy %d step %.3f exit %d" % (t["ell"], t["nnz"], t["nnz_xx"], t["nnz_yy"], t["step"], t["exit_code"]))
