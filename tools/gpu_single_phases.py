#!/usr/bin/env python3
"""One registration at a time (cvo_hip_align), per phase of the length-scale schedule: the registration is
stopped after 4 / 11 / 21 / all iterations (cvo_hip_params::max_iter; ell changes at the END of iterations
3, 10, 20, ref src/cvo.cpp:408-410) and the differences give us per iteration at ell = 0.15 / 0.10 / 0.06 / 0.03.
usage: gpu_single_phases.py [n] [reps] [cvo|acvo]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

pkg = ge.load_package(); capi = pkg.capi
if os.environ.get("CVO_LIB"):   # A/B of two builds in one session
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), os.environ["CVO_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
acvo = len(sys.argv) > 3 and sys.argv[3] == "acvo"
xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=pkg.data.SEED_CFG2, acvo=acvo)
mode = capi.MODE_ACVO if acvo else capi.MODE_CVO
res = []
for cap in (4, 11, 21, 0):
    p = capi.default_params(mode)
    if cap:
        p.max_iter = cap
    c = capi.Context(mode=mode, device=0, stream=torch.cuda.current_stream().cuda_stream, params=p)
    c.set_fixed(xf, ff); c.set_moving(xm, fm)
    for _ in range(3):
        st = capi.init_state(c.params); n_it, _ = c.align(st, trace_cap=0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        st = capi.init_state(c.params); n_it, _ = c.align(st, trace_cap=0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    res.append((n_it, dt * 1e6))
    c.close()
print("single %s %d x %d: %.1f registrations/s, %.3f ms, %d iterations, %.2f us per iteration" % (
    "acvo" if acvo else "cvo", n, n, 1e6 / res[-1][1], res[-1][1] / 1e3, res[-1][0], res[-1][1] / res[-1][0]))
prev_it, prev_t = 0, 0.0
for (it, t), name in zip(res, ("0.15", "0.10", "0.06", "0.03")):
    if it > prev_it:
        print("  ell %s: iterations %d..%d  %.1f us  = %.2f us per iteration%s" % (
            name, prev_it, it - 1, t - prev_t, (t - prev_t) / (it - prev_it), "  (incl. the call's fixed cost)" if prev_it == 0 else ""))
    prev_it, prev_t = it, t
