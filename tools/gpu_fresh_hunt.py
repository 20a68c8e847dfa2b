#!/usr/bin/env python3
"""The final state's pinned copy (head_publish -> job_pump): COUNT distinct pairs, each registered on a FRESH context (whose
mirror holds zeros) LOOPS times, against the same registration with the state fetched by a copy in stream order
(CVO_HIP_NO_FINAL_MIRROR=1); mismatches and the library's count of re-read mirrors.
usage: gpu_fresh_hunt.py [n] [count] [loops]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 32
loops = int(sys.argv[3]) if len(sys.argv) > 3 else 20
pairs = [pkg.data.synthetic_pair(n, n, seed=pkg.data.SEED_CFG5_BASE + b) for b in range(count)]
def once(pr):
    c = capi.Context(mode=capi.MODE_CVO, device=0)
    c.set_fixed(pr[0], pr[1]); c.set_moving(pr[2], pr[3])
    st = capi.init_state(c.params); it, _ = c.align(st, trace_cap=0)
    c.close()
    return it, bytes(st)
os.environ["CVO_HIP_NO_FINAL_MIRROR"] = "1"
ref = [once(pr) for pr in pairs]
os.environ.pop("CVO_HIP_NO_FINAL_MIRROR")
bad = 0; t0 = time.time()
for loop in range(loops):
    for b, pr in enumerate(pairs):
        got = once(pr)
        if got != ref[b]:
            bad += 1
            a = np.frombuffer(ref[b][1], np.uint32, 62); g = np.frombuffer(got[1], np.uint32, 62)
            print("MISMATCH loop %d pair %d: iterations %d vs %d; words %s" % (loop, b, got[0], ref[b][0],
                  ", ".join("%d: %08x/%08x" % (q, g[q], a[q]) for q in range(62) if a[q] != g[q])[:300]), flush=True)
print("n %d: %d fresh registrations, %d mismatches, mirror re-reads %d, %.0f s" % (n, loops * count, bad, capi.mirror_retries(), time.time() - t0))
