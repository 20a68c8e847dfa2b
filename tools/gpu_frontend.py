#!/usr/bin/env python3
"""Front end timing: frames/s of create_pointcloud (host buffers in, cloud out) and the
CPU restatement beside it.  usage: gpu_frontend.py [frames] [texture]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
from oracle import pyoracle_fe as fo

pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tex = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
frames = [pkg.data.synthetic_rgbd_frame(seed=100 + k, texture=tex) for k in range(8)]
gen = pkg.frontend.PcdGenerator(640, 480)
for bgr, dep in frames:
    gen.create_pointcloud(bgr, dep)
t0 = time.perf_counter()
for k in range(n):
    bgr, dep = frames[k % 8]
    xyz, feat = gen.create_pointcloud(bgr, dep)
dt = (time.perf_counter() - t0) / n
print("front end: %.3f ms per frame (%.0f frames/s), %d points, info %s" % (dt * 1e3, 1 / dt, len(xyz), gen.info()))
img, dep = gen.host_buffers()
img[...] = frames[0][0]; dep[...] = frames[0][1]
t0 = time.perf_counter()
for k in range(n):
    gen.create_pointcloud(img, dep)
dz = (time.perf_counter() - t0) / n
print("front end, frame already in the staging images: %.3f ms per frame (%.0f frames/s)" % (dz * 1e3, 1 / dz))
t0 = time.perf_counter()
for k in range(8):
    fo.create_pointcloud(*frames[k])
dc = (time.perf_counter() - t0) / 8
print("cpu restatement: %.2f ms per frame (1 thread)  ->  x%.1f" % (dc * 1e3, dc / dt))
