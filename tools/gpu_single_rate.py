#!/usr/bin/env python3
"""One registration at a time (cvo_hip_align) over a few sizes and seeds: registrations/s, iterations, run statistics.
usage: gpu_single_rate.py [n ...]   (env: SEEDS=a,b,c REPS=30)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

pkg = ge.load_package(); capi = pkg.capi
if os.environ.get("CVO_LIB"):
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), os.environ["CVO_LIB"])
sizes = [int(a) for a in sys.argv[1:]] or [3000, 6000, 10000]
seeds = [int(s) for s in os.environ.get("SEEDS", "%d,1001,1002" % pkg.data.SEED_CFG2).split(",")]
reps = int(os.environ.get("REPS", "30"))
acvo = bool(os.environ.get("ACVO"))
mode = capi.MODE_ACVO if acvo else capi.MODE_CVO
for n in sizes:
    for seed in seeds:
        xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=seed, acvo=acvo)
        c = capi.Context(mode=mode, device=0)
        c.set_fixed(xf, ff); c.set_moving(xm, fm)
        for _ in range(3):
            st = capi.init_state(c.params); n_it, _ = c.align(st, trace_cap=0)
        r0 = c.run_stats()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps):
            st = capi.init_state(c.params); n_it, _ = c.align(st, trace_cap=0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
        r1 = c.run_stats()
        print("n %5d seed %8d: %7.1f /s  %.3f ms  %3d iterations (%.2f us each); last registration: runs %d declined %d iterations inside %d; state %s" % (
            n, seed, 1.0 / dt, dt * 1e3, n_it, dt * 1e6 / n_it, r1[0], r1[1], r1[2],
            __import__("hashlib").sha1(bytes(st)).hexdigest()[:10]))
        c.close()
