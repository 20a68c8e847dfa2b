#!/usr/bin/env python3
"""Cost of handing a cloud over: set_moving from host arrays and from device arrays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
for n in (10000, 3000, 1000, 3000, 50000, 200000):
    xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=5)
    c = capi.Context(mode=capi.MODE_CVO, device=0)
    c.set_fixed(xf, ff)
    dx, df = torch.from_numpy(xm).cuda(), torch.from_numpy(fm).cuda()
    torch.cuda.synchronize()
    for _ in range(3):
        c.set_moving(xm, fm); c.set_moving_device(dx.data_ptr(), df.data_ptr(), n)
    reps = 50
    t = time.perf_counter()
    for _ in range(reps): c.set_moving(xm, fm)
    th = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for _ in range(reps): c.set_moving_device(dx.data_ptr(), df.data_ptr(), n)
    td = (time.perf_counter() - t) / reps
    print("n %6d: set_moving (host arrays) %.3f ms, set_moving_device %.3f ms" % (n, th * 1e3, td * 1e3))
    c.close()
