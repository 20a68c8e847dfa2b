#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CVO_HIP_GRAPH=1
echo "== engine debug, 64 distinct"; CVO_HIP_ENGINE_DEBUG=1 DISTINCT=1 timeout 300 python tools/gpu_batch.py 10000 3 64 2>&1 | grep -v amdgpu.ids | tail -30
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, time
import __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
ctxs=[]
for b in range(64):
    s = torch.cuda.Stream()
    xf, ff, xm, fm = pkg.data.synthetic_pair(10000, 10000, seed=1000+b)
    c = capi.Context(mode=capi.MODE_CVO, device=0, stream=s.cuda_stream, graph_capture=True)
    c.set_fixed(xf, ff); c.set_moving(xm, fm); ctxs.append((c,s))
cs=[c for c,_ in ctxs]
its = capi.align_many(cs, [capi.init_state(c.params) for c in cs])
print("iterations sorted:", sorted(its))
for cap in (2000, 120, 100, 80, 60):
    for c in cs:
        p = c.params; p.max_iter = cap; c.set_params(p)
    capi.align_many(cs, [capi.init_state(c.params) for c in cs])
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(5):
        its = capi.align_many(cs, [capi.init_state(c.params) for c in cs])
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
    print("max_iter %4d: %.2f ms per step, %d iterations in all, %.2f us per registration-iteration" % (cap, dt*1e3, sum(its), dt*1e6/sum(its)))
PY
