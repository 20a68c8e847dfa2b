#!/usr/bin/env python3
"""tools/gpu_small_calls.py in a process that has run a large cvo_hip_align_many call first (the engines' library-owned streams exist, as in
bench.py): do the few-registration calls suffer from the streams that are about?  usage: gpu_small_calls_after_engines.py [k ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
big = []
for i in range(64):
    xf, ff, xm, fm = pkg.data.synthetic_pair(3000, 3000, seed=pkg.data.SEED_CFG5_BASE + 500 + i)
    s = torch.cuda.Stream(); c = capi.Context(mode=capi.MODE_CVO, device=0, stream=s.cuda_stream, graph_capture=True)
    c.set_fixed(xf, ff); c.set_moving(xm, fm); big.append((c, s))
for _ in range(2): capi.align_many([c for c, _ in big], [capi.init_state(c.params) for c, _ in big])
if not os.environ.get("KEEP_BIG"):
    for c, _ in big: c.close()
    big = []
torch.cuda.synchronize()
for k in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
    cs, ss = [], []
    for i in range(k):
        xf, ff, xm, fm = pkg.data.synthetic_pair(3000, 3000, seed=pkg.data.SEED_CFG5_BASE + 100 + i)
        s = torch.cuda.Stream(); c = capi.Context(mode=capi.MODE_CVO, device=0, stream=s.cuda_stream, graph_capture=True)
        c.set_fixed(xf, ff); c.set_moving(xm, fm); cs.append(c); ss.append(s)
    for _ in range(3): capi.align_many(cs, [capi.init_state(c.params) for c in cs])
    torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 16
    for _ in range(reps): capi.align_many(cs, [capi.init_state(c.params) for c in cs])
    torch.cuda.synchronize(); r = reps * k / (time.perf_counter() - t0)
    print("after the engines: %2d per call on their own %7.1f /s, entries given up %d, runs %s" % (k, r, sum(c.get_option("run_aborts") for c in cs), [c.run_stats()[:2] for c in cs]))
    for c in cs: c.close()
