#!/usr/bin/env python3
"""Fixed cost of one cvo_hip_align() call: registrations that stop after 1, 2, 5 iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
xf, ff, xm, fm = pkg.data.synthetic_pair(10000, 10000, seed=pkg.data.SEED_CFG2)
for mi in (1, 2, 5, 9, 17):
    prm = capi.default_params(capi.MODE_CVO); prm.max_iter = mi
    c = capi.Context(mode=capi.MODE_CVO, device=0, stream=torch.cuda.current_stream().cuda_stream, params=prm)
    c.set_fixed(xf, ff); c.set_moving(xm, fm)
    for _ in range(3):
        st = capi.init_state(c.params); c.align(st, trace_cap=0)
    torch.cuda.synchronize(); t = time.perf_counter()
    reps = 50
    for _ in range(reps):
        st = capi.init_state(c.params); n_it, _ = c.align(st, trace_cap=0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    print("max_iter %2d: %d iterations, %.1f us per align()" % (mi, n_it, dt * 1e6))
    c.close()
