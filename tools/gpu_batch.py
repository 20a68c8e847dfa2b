#!/usr/bin/env python3
"""Throughput probe: B independent registrations in flight through
cvo_hip_align_many (one context + one HIP stream each, one host thread)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
capi = pkg.capi
if os.environ.get("CVO_LIB"):   # A/B of two builds in one session: a second library next to the built one
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), os.environ["CVO_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
BS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8, 16]
ACVO = len(sys.argv) > 4 and sys.argv[4] == "acvo"
IDLE = int(os.environ.get("IDLE_STREAMS", "0"))   # experiment: streams that exist but are never used
idle = [torch.cuda.Stream() for _ in range(IDLE)]
for B in BS:
    ctxs, streams = [], []
    for b in range(B):
        s = torch.cuda.Stream()
        xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=(1000 + b) if os.environ.get("DISTINCT") else pkg.data.SEED_CFG2, acvo=ACVO)
        c = capi.Context(mode=capi.MODE_ACVO if ACVO else capi.MODE_CVO, device=0, stream=s.cuda_stream)
        if os.environ.get("MAX_ITER"):   # probe: stop every registration after that many iterations
            prm = c.params; prm.max_iter = int(os.environ["MAX_ITER"])
            if os.environ.get("NO_BREAK"):   # ... and never earlier (probes whose sums are not the real ones)
                prm.eps = 0.0; prm.eps_2 = 0.0
            c.set_params(prm)
        c.set_fixed(xf, ff); c.set_moving(xm, fm)
        ctxs.append(c); streams.append(s)
    def step():
        t0 = time.perf_counter()
        states = [capi.init_state(c.params) for c in ctxs]
        t1 = time.perf_counter()
        r = capi.align_many(ctxs, states)
        if os.environ.get("PER_STEP"):
            print("   init_state %.2f ms, align_many %.2f ms" % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
        return r, states
    its, _ = step()
    torch.cuda.synchronize()
    import gc
    gc.collect(); gc.disable()   # (a full collector pass inside a step is a 35 ms pause of this probe)
    t = time.perf_counter()
    per = []
    for _ in range(reps):
        t1 = time.perf_counter()
        its, states = step()
        per.append((time.perf_counter() - t1) * 1e3)   # (align_many returns with every registration finished)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    gc.enable()
    if os.environ.get("PER_STEP"):
        print("   per step ms:", " ".join("%.2f" % x for x in per))
    import hashlib
    dig = hashlib.sha1(b"".join(bytes(st) for st in states) + bytes(its)).hexdigest()[:12]   # (equal across variants: same results)
    print("B %2d: %.1f registrations/s (%.2f ms per batch, iters %s, mean %.1f, %.2f us per registration-iteration) states %s" % (B, B * reps / dt, dt * 1e3 / reps, its[:3], float(np.mean(its)), dt * 1e6 / reps / max(1, int(np.sum(its))), dig))
    for c in ctxs: c.close()
