#!/usr/bin/env python3
"""Stress: T host threads, each with its own registration context and front end object,
R rounds each; every result must equal the thread's first one.  usage: gpu_threads.py [T] [R]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = []
def work(k):
    acvo = k % 2 == 1
    xf, ff, xm, fm = pkg.data.synthetic_pair(1200 + 150 * k, 1100 + 90 * k, seed=700 + k, acvo=acvo)
    bgr, dep = pkg.data.synthetic_rgbd_frame(width=320, height=256, seed=800 + k, texture=1.0)
    first = None
    for r in range(R):
        c = capi.Context(mode=capi.MODE_ACVO if acvo else capi.MODE_CVO, device=0) if r % 5 == 0 else c
        gen = pkg.frontend.PcdGenerator(320, 256) if r % 7 == 0 else gen
        c.set_fixed(xf, ff); c.set_moving(xm, fm)
        st = capi.init_state(c.params)
        it, _ = c.align(st, trace_cap=0)
        xyz, feat = gen.create_pointcloud(bgr, dep)
        res = (it, bytes(st), xyz.tobytes(), feat.tobytes())
        if first is None: first = res
        elif res != first: bad.append((k, r))
t0 = time.time()
ts = [threading.Thread(target=work, args=(k,)) for k in range(T)]
for t in ts: t.start()
for t in ts: t.join()
print("threads: %d x %d rounds, %d results differ, %.1f s" % (T, R, len(bad), time.time() - t0))
sys.exit(1 if bad else 0)
