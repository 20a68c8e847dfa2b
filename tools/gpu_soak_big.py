#!/usr/bin/env python3
"""Parity soak at cloud sizes above 65 536 rows (8-byte kept entries with 18-bit indices, ProcessArgs::kept_packed
== 2): random sizes, both modes, the first `max_iter` iterations against the oracle, bit for bit.
usage: gpu_soak_big.py [n_cases] [max_iter]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge
from oracle import pyoracle as po

pkg = ge.load_package(); capi = pkg.capi
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
max_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "2024")))
po.set_threads(int(os.environ.get("SOAK_THREADS", "16")))
bad = 0; t0 = time.time()
for case in range(n_cases):
    acvo = bool(case & 1)
    n = int(rng.integers(66000, 150000)); m = int(rng.integers(66000, 150000))
    xf, ff, xm, fm = pkg.data.synthetic_pair(n, m, seed=int(rng.integers(1, 10**6)), acvo=acvo)
    mode = capi.MODE_ACVO if acvo else capi.MODE_CVO
    prm = capi.default_params(mode); prm.max_iter = max_iter
    c = capi.Context(mode=mode, device=0, stream=torch.cuda.current_stream().cuda_stream, params=prm)
    c.set_fixed(xf, ff); c.set_moving(xm, fm)
    st = capi.init_state(c.params)
    it, tr = c.align(st, trace_cap=max_iter)
    c.close()
    p = po.default_params(po.MODE_ACVO if acvo else po.MODE_CVO); p.max_iter = max_iter
    so = po.init_state(p)
    n_or, tr_or = po.align(p, so, xf, ff, xm, fm, search=po.SEARCH_GRID)
    ok = it == n_or
    for a, b in zip(tr, tr_or):
        ok = ok and a["nnz"] == b["nnz"] and a["omega"] == b["omega"] and a["v"] == b["v"] and a["step"] == b["step"]
    R_ok = bytes(st.R) == bytes(so.R) and bytes(st.T) == bytes(so.T)
    bad += not (ok and R_ok)
    print("case %d %s %d x %d: %s (%d iterations, nnz %d ... %d)" % (case, "acvo" if acvo else "cvo", n, m,
          "ok" if ok and R_ok else "MISMATCH", it, tr[0]["nnz"], tr[-1]["nnz"]), flush=True)
print("big soak: %d cases, %d mismatches vs oracle, %.0f s" % (n_cases, bad, time.time() - t0))
