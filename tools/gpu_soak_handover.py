#!/usr/bin/env python3
"""Soak of the one-launch hand-over (k_cloud_one / cvo_hip_set_pcd_many) against the ten-launch preparation of
round 1-3 (CVO_HIP_NO_CLOUD_ONE): random clouds of 1 ... 16384 points -- uniform boxes, thin sheets, clumps of
duplicated points, huge offsets, a few non-finite coordinates --, both feature layouts, batches of random
composition (clouds above the one-launch limit mixed in).  Device arrays (packed rows, features, Morton order,
bounding spheres, padding rows) must be bit-identical.  usage: gpu_soak_handover.py [batches] [clouds per batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_package()
capi = pkg.capi
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "404")))
batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
per = int(sys.argv[2]) if len(sys.argv) > 2 else 6


def cloud():
    kind = rng.integers(0, 6)
    n = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 1000, 4097, 16383, 16384, int(rng.integers(1, 16385)), int(rng.integers(1, 3000)),
                        int(rng.integers(16385, 30000)) if rng.random() < 0.3 else int(rng.integers(1, 16385))]))
    x = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    if kind == 1:
        x[:, 2] = 1.3 + 1e-4 * x[:, 2]
    elif kind == 2:
        x = np.repeat(x[: max(1, n // 7)], 8, axis=0)[:n].copy()   # duplicated points: equal Morton keys
    elif kind == 3:
        x += np.float32(150.0)
    elif kind == 4 and n > 10:
        x[rng.integers(0, n, 3)] = np.float32("nan")
    elif kind == 5:
        x[:] = x[0]                                                 # one point n times: zero extent
    f = rng.uniform(0, 255, (n, 5)).astype(np.float32)
    return x, f


def arrays(c):
    out = []
    for which in (0, 1):
        d = c.device_cloud(which)
        out.append((d["rows"], d["points"], d["pos"].tobytes(), d["feat"].tobytes(), d["seg"].tobytes()))
    return out


bad = 0
for b in range(batches):
    col = bool(rng.integers(0, 2))
    lay = capi.FEAT_COLMAJOR if col else capi.FEAT_ROWMAJOR
    conv = (lambda a: np.ascontiguousarray(a.T)) if col else (lambda a: a)
    pairs = [(cloud(), cloud()) for _ in range(per)]
    os.environ["CVO_HIP_NO_CLOUD_ONE"] = "1"
    ref = []
    for (xf, ff), (xm, fm) in pairs:
        c = capi.Context(mode=capi.MODE_CVO, device=0)
        c.set_fixed(xf, conv(ff), layout=lay); c.set_moving(xm, conv(fm), layout=lay)
        ref.append(arrays(c))
        c.close()
    del os.environ["CVO_HIP_NO_CLOUD_ONE"]
    cs = [capi.Context(mode=capi.MODE_CVO, device=0) for _ in pairs]
    capi.set_pcd_many(cs, [(p[0][0], conv(p[0][1])) for p in pairs], [(p[1][0], conv(p[1][1])) for p in pairs], layout=lay)
    got = [arrays(c) for c in cs]
    # ... and once more into the same contexts, one call per cloud (the one-launch path of cvo_hip_set_*)
    for c, ((xf, ff), (xm, fm)) in zip(cs, pairs):
        c.set_moving(xm, conv(fm), layout=lay); c.set_fixed(xf, conv(ff), layout=lay)
    got2 = [arrays(c) for c in cs]
    for c in cs:
        c.close()
    for k in range(per):
        if got[k] != ref[k] or got2[k] != ref[k]:
            bad += 1
            print("MISMATCH batch %d cloud pair %d sizes %d %d (batched %s, per call %s)" % (b, k, len(pairs[k][0][0]), len(pairs[k][1][0]), got[k] == ref[k], got2[k] == ref[k]))
print("hand-over soak: %d batches x %d cloud pairs, %d mismatches" % (batches, per, bad))
sys.exit(1 if bad else 0)
