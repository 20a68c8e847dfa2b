#!/usr/bin/env python3
"""Repro loop: `world` ranks on one GPU through device mailboxes (tests/helpers.py align_two_ranks), many times in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import __graft_entry__ as ge
from helpers import align_two_ranks
pkg = ge.load_package(); capi = pkg.capi
world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
warm = len(sys.argv) > 4 and sys.argv[4] == "warm"
xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=61)
if warm:   # something else first: a lone registration and a small align_many (engines, their streams)
    c = capi.Context(mode=capi.MODE_CVO, device=0, stream=torch.cuda.current_stream().cuda_stream)
    c.set_fixed(xf, ff); c.set_moving(xm, fm); c.align(capi.init_state(c.params), trace_cap=0); c.close()
    cs = []
    for i in range(12):
        a = pkg.data.synthetic_pair(1500, 1500, seed=500 + i)
        s = torch.cuda.Stream(); cc = capi.Context(mode=capi.MODE_CVO, device=0, stream=s.cuda_stream); cc.set_fixed(a[0], a[1]); cc.set_moving(a[2], a[3]); cs.append((cc, s))
    capi.align_many([q[0] for q in cs], [capi.init_state(q[0].params) for q in cs])
    for q in cs: q[0].close()
ok = bad = 0
for k in range(reps):
    t = time.time()
    try:
        out = align_two_ranks(pkg, capi.MODE_CVO, xf, ff, xm, fm, exchange="mailbox", world=world, timeout=60)
        same = all(out[r][1] == out[0][1] for r in range(world))
        ok += 1 if same else 0; bad += 0 if same else 1
        print("rep %d: %d iterations, lock step %s, %.2f s" % (k, out[0][0], same, time.time() - t), flush=True)
    except AssertionError as e:
        bad += 1
        print("rep %d: FAILED %.2f s: %s" % (k, time.time() - t, str(e)[:160]), flush=True)
print("world %d n %d: %d ok, %d bad" % (world, n, ok, bad))
