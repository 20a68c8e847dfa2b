#!/usr/bin/env python3
"""Quick GPU timing probe: one cvo align() on a synthetic N x N pair."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
mode = sys.argv[2] if len(sys.argv) > 2 else "cvo"
acvo = mode == "acvo"
xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=pkg.data.SEED_CFG2, acvo=acvo)
Reg = pkg.Acvo if acvo else pkg.Cvo
for rep in range(3):
    reg = Reg(device=0, stream=torch.cuda.current_stream().cuda_stream)
    reg.ctx.set_profiling(rep == 2)
    reg.run_cvo(xf, ff)
    torch.cuda.synchronize()
    t = time.time()
    reg.run_cvo(xm, fm, trace_cap=2000)
    torch.cuda.synchronize()
    dt = time.time() - t
    print("rep", rep, "n", n, mode, "iters", reg.num_iterations, "total ms %.2f" % (dt * 1e3),
          "ms/iter %.3f" % (dt * 1e3 / reg.num_iterations))
    if rep == 2:
        pr = reg.ctx.get_profile()
        print(pr)
        for k in ("flow", "step", "self"):
            if pr[k + "_launches"]:
                ms = pr[k + "_ms"] / pr[k + "_launches"]
                pairs = pr[k + "_pairs"] / pr[k + "_launches"]
                print(k, "avg ms/launch %.4f" % ms, "Gpairs/s %.1f" % (pairs / ms / 1e6),
                      "TFLOP/s(8 flop/pair) %.2f" % (8 * pairs / ms / 1e9))
        print("nnz per iteration:", [t_["nnz"] for t_ in reg.trace][:30])
    rot, tr = pkg.data.rel_pose_error(np.linalg.inv(reg.transform.astype(np.float64)), np.linalg.inv(pkg.data.gt_motion()))
    reg.close()
print("final transform vs ground truth motion (rel rot err, rel trans err):", rot, tr)

# cloud hand-over cost (host Morton sort + H2D), the PCIe-inclusive part of a frame
ctx = pkg.capi.Context(mode=pkg.capi.MODE_CVO, device=0, stream=torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    t = time.time()
    ctx.set_moving(xm, fm)
    ctx.synchronize()
    dt = time.time() - t
print("set_moving (%d points, sort + upload): %.3f ms" % (n, dt * 1e3))
ctx.close()
