import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import pyoracle as po
pkg = ge.load_package()
xf, ff, xm, fm = pkg.data.synthetic_pair(2000, 2000, seed=7, acvo=True)
reg = pkg.Acvo(device=0, stream=torch.cuda.current_stream().cuda_stream)
reg.run_cvo(xf, ff); reg.run_cvo(xm, fm, trace_cap=2000)
p = po.default_params(1); st = po.init_state(p)
n, tr = po.align(p, st, xf, ff, xm, fm)
print(n, reg.num_iterations)
print(np.array(st.R).reshape(3,3)); print(np.array(reg.state.R).reshape(3,3))
print(np.array(st.T), np.array(reg.state.T))
for a, b in zip(reg.trace, tr):
    if a != b:
        d = {k: (a[k], b[k]) for k in a if a[k] != b[k] and not (isinstance(a[k], float) and a[k] != a[k])}
        if d: print(a['k'], d); break
print(reg.trace[-1]); print(tr[-1])
