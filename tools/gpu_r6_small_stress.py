#!/usr/bin/env python3
"""Small cvo_hip_align_many calls over and over (2, 4, 8 registrations of 3000 points a call): entries that gave up at the hand-shake
(run_aborts), and whether any call returned other bytes than the first.  usage: gpu_r6_small_stress.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
big = [pkg.data.synthetic_pair(3000, 3000, seed=8900 + b) for b in range(8)]
ref = {}
tot = {2: [0, 0, 0], 4: [0, 0, 0], 8: [0, 0, 0]}   # calls, aborts, differing results
for r in range(rounds):
    for k in (2, 4, 8):
        cs, ss = [], []
        for xf, ff, xm, fm in big[:k]:
            s = torch.cuda.Stream()
            c = capi.Context(mode=capi.MODE_CVO, device=0, stream=s.cuda_stream)
            c.set_fixed(xf, ff); c.set_moving(xm, fm)
            cs.append(c); ss.append(s)
        for _ in range(4):
            states = [capi.init_state(c.params) for c in cs]
            its = capi.align_many(cs, states)
            got = [(i, bytes(s)) for i, s in zip(its, states)]
            for b, g in enumerate(got):
                if ref.setdefault(b, g) != g:
                    tot[k][2] += 1
            tot[k][0] += 1
        tot[k][1] += int(sum(c.get_option("run_aborts") for c in cs))
        for c in cs:
            c.close()
for k in (2, 4, 8):
    print("k %d: %d calls, %d entries given up, %d results that differ" % (k, *tot[k]))
