#!/usr/bin/env python3
"""How evenly one flow pass spreads its work over blocks / XCD regions (diagnostics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=pkg.data.SEED_CFG2)
c = pkg.capi.Context(mode=pkg.capi.MODE_CVO, device=0)
c.set_fixed(xf, ff); c.set_moving(xm, fm)
c.transform_pcd(np.eye(3, dtype=np.float32), np.zeros(3, np.float32))
for ell in (0.15, 0.10, 0.06, 0.03):
    out = c.flow(ell)
    w = c.wave_load()
    b = w.reshape(-1, 4).sum(1)
    reg = np.array([b[r::8].sum() for r in range(8)])
    print("ell %.2f nnz %d: per wave mean %.1f max %d (max/mean %.2f); per block mean %.1f max %d min %d (max/mean %.2f); per XCD region max/mean %.2f %s"
          % (ell, int(out[8]), w.mean(), w.max(), w.max() / max(w.mean(), 1e-9), b.mean(), b.max(), b.min(), b.max() / max(b.mean(), 1e-9),
             reg.max() / reg.mean(), list(reg)))
c.close()
