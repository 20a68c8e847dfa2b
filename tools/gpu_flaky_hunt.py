#!/usr/bin/env python3
"""Hunt for an intermittent mismatch: COUNT distinct pairs, the reference state of each from a registration on its own without
resident runs (CVO_HIP_NO_RUN contexts); then LOOPS times: all pairs through the engines (cvo_hip_align_many), then each on its
own (cvo_hip_align) on the same contexts -- every state against the reference, differing floats printed.
usage: gpu_flaky_hunt.py [n] [count] [loops]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 64
loops = int(sys.argv[3]) if len(sys.argv) > 3 else 10
pairs = [pkg.data.synthetic_pair(n, n, seed=(pkg.data.SEED_CFG2 if b == 0 else pkg.data.SEED_CFG5_BASE + b)) for b in range(count)]
os.environ["CVO_HIP_NO_RUN"] = "1"
ref = []
for pr in pairs:
    c = capi.Context(mode=capi.MODE_CVO, device=0)
    c.set_fixed(pr[0], pr[1]); c.set_moving(pr[2], pr[3])
    st = capi.init_state(c.params); it, _ = c.align(st, trace_cap=0)
    ref.append((it, bytes(st))); c.close()
os.environ.pop("CVO_HIP_NO_RUN")
ctxs, streams = [], []
for pr in pairs:
    s = torch.cuda.Stream()
    c = capi.Context(mode=capi.MODE_CVO, device=0, stream=s.cuda_stream, graph_capture=True)
    c.set_fixed(pr[0], pr[1]); c.set_moving(pr[2], pr[3])
    ctxs.append(c); streams.append(s)
def report(kind, loop, b, it, got):
    r_it, r = ref[b]
    a = np.frombuffer(r, np.float32, 62); g = np.frombuffer(got, np.float32, 62)
    d = [q for q in range(62) if a[q].view(np.uint32) != g[q].view(np.uint32)]
    print("MISMATCH %s loop %d pair %d: iterations %d vs %d; floats %s" % (kind, loop, b, it, r_it,
          ", ".join("%d: %08x/%08x" % (q, int(g[q].view(np.uint32)), int(a[q].view(np.uint32))) for q in d[:8])), flush=True)
bad = 0
t0 = time.time()
for loop in range(loops):
    states = [capi.init_state(c.params) for c in ctxs]
    its = capi.align_many(ctxs, states)
    for b in range(count):
        if (its[b], bytes(states[b])) != ref[b]: bad += 1; report("engines", loop, b, its[b], bytes(states[b]))
    if os.environ.get("ENGINES_ONLY"): continue
    for b, c in enumerate(ctxs):
        st = capi.init_state(c.params); it, _ = c.align(st, trace_cap=0)
        if (it, bytes(st)) != ref[b]: bad += 1; report("alone", loop, b, it, bytes(st))
print("loops %d x %d pairs: %d mismatches, %.0f s" % (loops, count, bad, time.time() - t0))
