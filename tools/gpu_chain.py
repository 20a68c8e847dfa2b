#!/usr/bin/env python3
"""Long chain parity: a synthetic RGB-D sequence through the front end and ONE registration
object (state carried from pair to pair as in the reference's drivers), against the oracle
chain on the oracle's clouds.  usage: gpu_chain.py [frames] [cvo|acvo]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
from oracle import pyoracle as po, pyoracle_fe as fo
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
acvo = len(sys.argv) > 2 and sys.argv[2] == "acvo"
po.set_threads(16)
frames = [pkg.data.synthetic_rgbd_frame(seed=55, texture=1.0 + 0.5 * np.sin(k / 5.0), motion=(1.2 * k, 0.6 * np.sin(k / 3.0) * 4))
          for k in range(n)]
ftype = 0 if acvo else 1
reg = (pkg.Acvo if acvo else pkg.Cvo)()
gen = pkg.frontend.PcdGenerator(640, 480)
p = po.default_params(po.MODE_ACVO if acvo else po.MODE_CVO)
s = po.init_state(p)
prev = None
bad = 0
t0 = time.time()
for k, (bgr, dep) in enumerate(frames):
    gen.submit(bgr, dep, 1, ftype)
    d_xyz, d_feat, npts = gen.collect_device()
    reg.run_cvo_device(d_xyz, d_feat, npts)
    r = fo.create_pointcloud(bgr, dep, 1, ftype)
    cur = (r["positions"], r["features"])
    if prev is not None:
        if acvo:   # tail of acvo::set_pcd (ref src/adaptive_cvo.cpp:476-478)
            s.ell = p.ell_init; s.ell_max = p.ell_max_init
        it, _ = po.align(p, s, prev[0], prev[1], cur[0], cur[1], search=po.SEARCH_GRID, trace_cap=0)
        T_or, _, A_or = po.state_matrices(s)
        same = (it == reg.num_iterations and np.array_equal(T_or, reg.transform) and np.array_equal(A_or, reg.accum_transform))
        if not same:
            bad += 1
            print("pair %d: gpu %d iterations, oracle %d; |dT| %.3g" % (k, reg.num_iterations, it, np.abs(T_or - reg.transform).max()))
    prev = cur
print("chain: %d frames (%s), %d pairs differ from the oracle chain, %.0f s" % (n, "acvo" if acvo else "cvo", bad, time.time() - t0))
sys.exit(1 if bad else 0)
