#!/usr/bin/env python3
"""`world` ranks on ONE GPU, one host thread and one library-owned stream each, through mailboxes in device memory
(tests/helpers.py align_two_ranks) against the unsharded registration: lock step bit for bit, the same iteration count, transform
within 1e-6.  A process of its own (tests/test_gpu_paths.py runs it as a subprocess): the ranks' streams are then the first streams of
the process and get hardware queues of their own -- inside a long-lived process the runtime hands a new stream the least-used
hardware queue, two ranks can end up on one, and a rank that spins in a kernel for its peer keeps that peer's kernels from starting
(both time out).  Ranks that share a GPU exist in tests only.
usage: gpu_ranks_threads.py cvo|acvo n m world [in_launch] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import __graft_entry__ as ge
from helpers import align_two_ranks

mode_name, n, m, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
if len(sys.argv) > 5 and sys.argv[5] == "in_launch":
    os.environ["CVO_HIP_TWIST_ON_SHARED_GPU"] = "1"
pkg = ge.load_package(); capi = pkg.capi
acvo = mode_name == "acvo"
mode = capi.MODE_ACVO if acvo else capi.MODE_CVO
seed = int(sys.argv[6]) if len(sys.argv) > 6 else 61
xf, ff, xm, fm = pkg.data.synthetic_pair(n, m, seed=seed, acvo=acvo)
out = align_two_ranks(pkg, mode, xf, ff, xm, fm, exchange="mailbox", world=world, timeout=120)
ref = capi.Context(mode=mode, device=0)
ref.set_fixed(xf, ff); ref.set_moving(xm, fm)
st_ref = capi.init_state(ref.params)
it_ref, _ = ref.align(st_ref, trace_cap=0)
ref.close()
T_ref = np.array(st_ref.transform, np.float32).reshape(4, 4)
for r in range(world):
    assert out[r][0] == it_ref, (r, out[r][0], it_ref)
    assert out[r][1] == out[0][1], "rank %d left lock step" % r
rot, tra = pkg.data.rel_pose_error(out[0][2], T_ref)
assert rot <= 1e-6 and tra <= 1e-6, (rot, tra)
print("ranks on one gpu, world %d, %s %d x %d: OK (%d iterations, rot %.2g, trans %.2g)" % (world, mode_name, n, m, it_ref, rot, tra))
