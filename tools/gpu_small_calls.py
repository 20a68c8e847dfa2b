#!/usr/bin/env python3
"""cvo_hip_align_many with a few registrations per call (registrations/s, on their own streams against through the engines) and the
runs that gave up at their entry.  usage: gpu_small_calls.py [n] [k ...]   (env ACVO=1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
ks = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4, 6, 8, 12]
acvo = bool(os.environ.get("ACVO")); mode = capi.MODE_ACVO if acvo else capi.MODE_CVO
for k in ks:
    res = []
    for alone in (1, 0):
        cs, ss = [], []
        for i in range(k):
            xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=pkg.data.SEED_CFG5_BASE + 100 + i, acvo=acvo)
            s = torch.cuda.Stream(); c = capi.Context(mode=mode, device=0, stream=s.cuda_stream, graph_capture=True)
            c.set_fixed(xf, ff); c.set_moving(xm, fm); cs.append(c); ss.append(s)
        cs[0].set_option("small_calls_alone", alone)
        for _ in range(3): capi.align_many(cs, [capi.init_state(c.params) for c in cs])
        torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 16
        for _ in range(reps): capi.align_many(cs, [capi.init_state(c.params) for c in cs])
        torch.cuda.synchronize(); res.append(reps * k / (time.perf_counter() - t0))
        ab = sum(c.get_option("run_aborts") for c in cs)
        for c in cs: c.close()
        if alone: ab_alone = ab
    print("n %d, %2d per call: on their own %7.1f /s (entries given up %d), through the engines %7.1f /s" % (n, k, res[0], ab_alone, res[1]))
