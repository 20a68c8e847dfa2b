#!/usr/bin/env python3
"""One cvo_hip_align_many call of 64 distinct 10k pairs: which registrations left the engines in the call's tail and what they did alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
n, B = 10000, 64
cs, ss = [], []
for b in range(B):
    s = torch.cuda.Stream(); xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=1000 + b)
    c = capi.Context(mode=capi.MODE_CVO, device=0, stream=s.cuda_stream, graph_capture=True)
    c.set_fixed(xf, ff); c.set_moving(xm, fm); cs.append(c); ss.append(s)
for rep in range(3):
    h0 = [c.get_option("tail_handovers") for c in cs]
    torch.cuda.synchronize(); t = time.perf_counter()
    its = capi.align_many(cs, [capi.init_state(c.params) for c in cs]); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    left = [(b, its[b], cs[b].run_stats()) for b in range(B) if cs[b].get_option("tail_handovers") > h0[b]]
    print("call %d: %.2f ms; left the engines: %s" % (rep, dt * 1e3, ", ".join("#%d %d its runs %s" % (b, i, r) for b, i, r in left)))
