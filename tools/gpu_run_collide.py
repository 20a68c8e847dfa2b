#!/usr/bin/env python3
"""Large resident runs of two registrations colliding on one GPU: two host threads, each registering its own 10k x 10k pair
over and over on its own context (a run of 248 solver blocks wants the whole GPU: the other's entry hand-shake times out and it
declines, or waits its turn).  Every result against the pair registered alone.
usage: gpu_run_collide.py [n] [reps] [threads]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
nthr = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pairs = [pkg.data.synthetic_pair(n, n, seed=pkg.data.SEED_CFG5_BASE + 7 * t) for t in range(nthr)]
ctxs, ref = [], []
for pr in pairs:
    c = capi.Context(mode=capi.MODE_CVO, device=0)
    c.set_fixed(pr[0], pr[1]); c.set_moving(pr[2], pr[3])
    st = capi.init_state(c.params); it, _ = c.align(st, trace_cap=0)
    ref.append((it, bytes(st))); ctxs.append(c)
bad = [0] * nthr; stats = [None] * nthr
def work(t):
    c = ctxs[t]
    for _ in range(reps):
        st = capi.init_state(c.params); it, _ = c.align(st, trace_cap=0)
        if (it, bytes(st)) != ref[t]: bad[t] += 1
    stats[t] = c.run_stats()
t0 = time.time()
ths = [threading.Thread(target=work, args=(t,)) for t in range(nthr)]
for th in ths: th.start()
for th in ths: th.join()
dt = time.time() - t0
print("%d threads x %d registrations of %d x %d: %d mismatches, %.1f registrations/s in all; last registration of each: %s" % (
    nthr, reps, n, n, sum(bad), nthr * reps / dt, ", ".join("runs %d declined %d inside %d" % s[:3] for s in stats)))
