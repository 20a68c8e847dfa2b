#!/usr/bin/env python3
"""Front end parity soak: random synthetic frames (texture, size, density wanted, depth
holes, camera, feature type), HIP front end vs the C restatement, clouds and selection maps
bit for bit.  usage: gpu_soak_fe.py [n_cases]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
from oracle import pyoracle_fe as fo

pkg = ge.load_package()
F = pkg.frontend
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "2024")))
sizes = [(640, 480), (320, 240), (352, 288), (160, 128), (800, 600), (96, 64), (333, 251), (65, 64), (127, 193), (64, 64), (1024, 64), (1920, 1080)]
gens = {}
bad = 0
stats = {"reselected": 0, "canny": 0, "points": 0}
t0 = time.time()
for case in range(n_cases):
    w, h = sizes[int(rng.integers(0, len(sizes)))]
    tex = float(rng.choice([0.0, 0.05, 0.2, 0.5, 1.0, 2.0, 4.0, 8.0]))
    want = int(rng.choice([1, 30, 300, 1000, 3000, 8000, w * h]))
    holes = float(rng.choice([0.0, 0.02, 0.3]))
    seq = int(rng.integers(0, 7))
    ftype = int(rng.integers(0, 2))
    bgr, dep = pkg.data.synthetic_rgbd_frame(width=w, height=h, seed=int(rng.integers(1, 10**6)), texture=tex,
                                             motion=(float(rng.normal() * 3), float(rng.normal() * 3)), holes=holes)
    if rng.random() < 0.15:   # pure noise / saturated patches
        bgr = rng.integers(0, 256, bgr.shape, dtype=np.uint8)
    if rng.random() < 0.15:
        bgr[:h // 3] = 255; bgr[-h // 4:] = 0
    key = (w, h)
    if key not in gens:
        gens[key] = F.PcdGenerator(w, h)
    g = gens[key]
    g._chk(F.lib().cvo_fe_set_num_want(g._h, want), "set_num_want")
    xyz, feat = g.create_pointcloud(bgr, dep, seq, ftype)
    ref = fo.create_pointcloud(bgr, dep, seq, ftype, want)
    info = g.info()
    ok = (np.array_equal(xyz.view(np.uint32), ref["positions"].view(np.uint32)) and
          np.array_equal(feat.view(np.uint32), ref["features"].view(np.uint32)) and
          np.array_equal(g.read_stage(F.STAGE_MAP), ref["map"]) and info["num_selected"] == ref["num_selected"])
    stats["reselected"] += info["reselected"]; stats["canny"] += info["canny_used"]; stats["points"] += len(xyz)
    if not ok:
        bad += 1
        print("MISMATCH case %d: %dx%d texture %.2f want %d holes %.2f seq %d ftype %d: %d vs %d points" % (
            case, w, h, tex, want, holes, seq, ftype, len(xyz), len(ref["positions"])))
print("front end soak: %d cases, %d mismatches vs oracle, %d re-selected, %d with edge top-up, %.0f points avg, %.0f s" % (
    n_cases, bad, stats["reselected"], stats["canny"], stats["points"] / max(n_cases, 1), time.time() - t0))
sys.exit(1 if bad else 0)
