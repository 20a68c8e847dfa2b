#!/usr/bin/env python3
"""Fixed cost of one cvo_hip_align call: wall time against max_iter (the slope is the iteration, the intercept the call).
usage: gpu_call_overhead.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_package(); capi = pkg.capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=pkg.data.SEED_CFG2)
pts = []
for mi in (1, 2, 4, 8, 16, 24, 32, 48):
    prm = capi.default_params(capi.MODE_CVO); prm.max_iter = mi
    c = capi.Context(mode=capi.MODE_CVO, device=0, stream=torch.cuda.current_stream().cuda_stream, params=prm)
    c.set_fixed(xf, ff); c.set_moving(xm, fm)
    for _ in range(5):
        st = capi.init_state(c.params); c.align(st, trace_cap=0)
    torch.cuda.synchronize(); t = time.perf_counter()
    reps = 60
    for _ in range(reps):
        st = capi.init_state(c.params); it, _ = c.align(st, trace_cap=0)
    dt = (time.perf_counter() - t) / reps
    pts.append((it, dt * 1e6))
    print("max_iter %3d: %d iterations, %.1f us per call" % (mi, it, dt * 1e6))
    c.close()
x = np.array([p[0] for p in pts[3:]], float); y = np.array([p[1] for p in pts[3:]])
k, b = np.polyfit(x, y, 1)
print("fit over max_iter >= 8: %.2f us per iteration + %.1f us per call" % (k, b))
