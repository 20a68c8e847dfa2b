#!/usr/bin/env python3
"""End-to-end stream probe: synthetic VGA frames -> front end -> run_cvo -> pose, with and
without prefetching the next frame.  usage: gpu_stream.py [frames] [cvo|acvo]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
mode = sys.argv[2] if len(sys.argv) > 2 else "cvo"
cls = pkg.Acvo if mode == "acvo" else pkg.Cvo
base = [("%d" % k,) + pkg.data.synthetic_rgbd_frame(seed=77, texture=1.0, motion=(1.5 * k, 0.7 * k)) for k in range(12)]
seq = [base[(k % 22) if (k % 22) < 12 else 22 - (k % 22)] for k in range(n)]   # forth and back
gen = pkg.frontend.PcdGenerator(640, 480)
for rep in range(3):
  for DEV in (False, True):
    for prefetch in (True, False):
        reg = cls()
        pkg.frontend.run_frames(reg, seq[:3], 1, generator=gen, prefetch=prefetch, device=DEV)
        reg.close()
        reg = cls()
        t0 = time.perf_counter()
        pkg.frontend.run_frames(reg, seq, 1, generator=gen, prefetch=prefetch, device=DEV)
        dt = (time.perf_counter() - t0) / len(seq)
        print("%s prefetch=%d device=%s: %.3f ms per frame (%.0f frames/s), last pair %d iterations" % (
            mode, prefetch, DEV, dt * 1e3, 1 / dt, reg.num_iterations))
        reg.close()
