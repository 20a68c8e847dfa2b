#!/usr/bin/env python3
"""Resident runs (kt_run) against the classic head-mode launches of the same library (CVO_HIP_NO_RUN=1, read when a
context is created): same iteration count, bit-identical final state and float32 trace, float64 sums to 1e-11; timing.
usage: gpu_r5_run_check.py [sizes...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_package(); capi = pkg.capi
sizes = [int(a) for a in sys.argv[1:]] or [2000, 3000, 6000, 10000]
bad = 0


def one(n, seed, no_run, trace_cap, reps):
    if no_run: os.environ["CVO_HIP_NO_RUN"] = "1"
    else: os.environ.pop("CVO_HIP_NO_RUN", None)
    xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=seed)
    c = capi.Context(mode=capi.MODE_CVO, device=0, stream=torch.cuda.current_stream().cuda_stream)
    c.set_fixed(xf, ff); c.set_moving(xm, fm)
    st = capi.init_state(c.params); n_it, tr = c.align(st, trace_cap=trace_cap)
    stats = c.run_stats()
    dt = 0.0
    if reps:
        for _ in range(2):
            s2 = capi.init_state(c.params); c.align(s2, trace_cap=0)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps):
            s2 = capi.init_state(c.params); c.align(s2, trace_cap=0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    c.close()
    return n_it, bytes(st), tr, stats, dt


for n in sizes:
    for seed, trace_cap in ((pkg.data.SEED_CFG2, 0), (pkg.data.SEED_CFG2, 2000), (1001, 2000), (1002, 0)):
        reps = 20 if seed == pkg.data.SEED_CFG2 and trace_cap == 0 else 0
        a = one(n, seed, False, trace_cap, reps)
        b = one(n, seed, True, trace_cap, reps)
        ok = a[0] == b[0] and a[1] == b[1]
        why = "" if ok else " STATE/ITER MISMATCH (%d vs %d)" % (a[0], b[0])
        if trace_cap and a[0] == b[0]:
            ta, tb = a[2], b[2]
            for k in range(a[0]):
                for f in ("k", "exit_code", "ell", "step", "nnz"):
                    va, vb = ta[k][f], tb[k][f]
                    if va != vb and not (va != va and vb != vb): ok = False; why += " trace[%d].%s %r %r" % (k, f, va, vb); break
                if not np.array_equal(np.array(ta[k]["omega"]), np.array(tb[k]["omega"])) or not np.array_equal(np.array(ta[k]["v"]), np.array(tb[k]["v"])):
                    ok = False; why += " trace[%d] twist" % k
                da = np.array(list(ta[k]["omega_d"]) + list(ta[k]["v_d"]) + list(ta[k]["bcde"]) + [ta[k]["sum_a"]]); db = np.array(list(tb[k]["omega_d"]) + list(tb[k]["v_d"]) + list(tb[k]["bcde"]) + [tb[k]["sum_a"]])
                if not np.allclose(da, db, rtol=1e-11, atol=1e-13): ok = False; why += " trace[%d] f64 sums %g" % (k, np.max(np.abs(da - db)))
                if not ok: break
        bad += 0 if ok else 1
        print("n %5d seed %8d trace %4d: %3d iterations, runs entered %d declined %d iterations in runs %d (last record %d candidates); %s%s%s" % (
            n, seed, trace_cap, a[0], a[3][0], a[3][1], a[3][2], a[3][3], "OK" if ok else "MISMATCH", why,
            ("  | %.1f /s (%.2f us/it) with runs, %.1f /s (%.2f us/it) without" % (1 / a[4], a[4] * 1e6 / a[0], 1 / b[4], b[4] * 1e6 / b[0])) if reps else ""))
print("mismatches:", bad)
sys.exit(1 if bad else 0)
