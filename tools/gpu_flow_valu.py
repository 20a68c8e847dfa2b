#!/usr/bin/env python3
"""One flow pass + one step-coefficient pass per length-scale of the cvo schedule on the configs[1] pair
(identity pose): run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES ...` to get the VALU cost per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
if os.environ.get("CVO_LIB"):
    pkg.capi.LIB_PATH = os.path.join(os.path.dirname(pkg.capi.LIB_PATH), os.environ["CVO_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=pkg.data.SEED_CFG2)
c = pkg.capi.Context(mode=pkg.capi.MODE_CVO, device=0)
c.set_fixed(xf, ff); c.set_moving(xm, fm)
c.transform_pcd(np.eye(3, dtype=np.float32), np.zeros(3, np.float32))
for ell in (0.15, 0.10, 0.06, 0.03):
    out = c.flow(ell)
    om, v = out[0:3].astype(np.float32), out[3:6].astype(np.float32)
    c.step_coeffs(om, v, ell)
    print("ell %.2f nnz %d" % (ell, int(out[8])))
c.close()
