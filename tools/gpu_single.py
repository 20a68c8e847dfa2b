#!/usr/bin/env python3
"""One registration at a time (cvo_hip_align): registrations/s and us per iteration.
usage: gpu_single.py [n] [reps] [cvo|acvo]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

pkg = ge.load_package(); capi = pkg.capi
if os.environ.get("CVO_LIB"):   # A/B of two builds in one session
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), os.environ["CVO_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
acvo = len(sys.argv) > 3 and sys.argv[3] == "acvo"
xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=pkg.data.SEED_CFG2, acvo=acvo)
c = capi.Context(mode=capi.MODE_ACVO if acvo else capi.MODE_CVO, device=0, stream=torch.cuda.current_stream().cuda_stream)
c.set_fixed(xf, ff); c.set_moving(xm, fm)
for _ in range(3):
    st = capi.init_state(c.params); n_it, _ = c.align(st, trace_cap=0)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(reps):
    st = capi.init_state(c.params); n_it, _ = c.align(st, trace_cap=0)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
print("single %s %d x %d: %.1f registrations/s, %.3f ms per registration, %d iterations, %.2f us per iteration" % (
    "acvo" if acvo else "cvo", n, n, 1 / dt, dt * 1e3, n_it, dt * 1e6 / n_it))
c.close()
