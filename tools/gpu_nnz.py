import os, sys
sys.path.insert(0, "/root/repo")
import torch, __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
for n in (3000, 6000, 10000):
    xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=pkg.data.SEED_CFG2)
    c = capi.Context(mode=capi.MODE_CVO, device=0, stream=torch.cuda.current_stream().cuda_stream)
    c.set_fixed(xf, ff); c.set_moving(xm, fm)
    st = capi.init_state(c.params); n_it, tr = c.align(st, trace_cap=2000)
    print(n, n_it, [(k, tr[k]["nnz"]) for k in (0, 3, 4, 10, 11, 20, 21, 30, n_it - 1)], c.run_stats())
    c.close()
