#!/usr/bin/env python3
"""Where block 0 of the resident runs spends its time (a -DCVO_RUN_CLOCKS build: tools/build_variant.sh clk -DCVO_RUN_CLOCKS).
usage: CVO_LIB=libcvo_hip_clk.so gpu_run_clocks.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

pkg = ge.load_package(); capi = pkg.capi
if os.environ.get("CVO_LIB"):
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), os.environ["CVO_LIB"])
names = ("entry", "consts", "flow rounds", "wave sums", "barrier+block sum", "exch A", "twist", "step rounds", "step sums", "exch B", "pre-head", "head_post", "inverse+sync", "tail", "exit", "(of head_post: cubic+root)")
for n in [int(a) for a in sys.argv[1:]] or [3000, 10000]:
    acvo = bool(os.environ.get("ACVO"))
    MODE = capi.MODE_ACVO if acvo else capi.MODE_CVO
    xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=pkg.data.SEED_CFG2, acvo=acvo)
    prm = capi.default_params(MODE)
    if os.environ.get("MAX_ITER"):   # (21: the run at ell = 0.06 alone)
        prm.max_iter = int(os.environ["MAX_ITER"])
    c = capi.Context(mode=MODE, device=0, stream=torch.cuda.current_stream().cuda_stream, params=prm)
    c.set_fixed(xf, ff); c.set_moving(xm, fm)
    for _ in range(2):
        st = capi.init_state(c.params); n_it, _ = c.align(st, trace_cap=0)
    torch.cuda.synchronize(); t = time.perf_counter()
    st = capi.init_state(c.params); n_it, _ = c.align(st, trace_cap=0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    runs, declined, its, cand = c.run_stats()
    clk = c.run_clocks()
    print("n %d: %d iterations in %.1f us; %d runs (%d declined), %d iterations inside, last record %d candidates" % (n, n_it, dt * 1e6, runs, declined, its, cand))
    if its:
        tot = sum(clk)
        print("   ticks of the first solver block (2.4 per ns): %.0f per run-iteration" % ((tot - clk[0] - clk[14]) / its))
        print("   " + ", ".join("%s %.0f" % (nm, (v / runs) if nm in ("entry", "exit") else (v / its)) for nm, v in zip(names, clk)) + "  (entry / exit per run, the others per iteration)")
    c.close()
