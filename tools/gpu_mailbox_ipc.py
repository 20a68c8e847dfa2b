#!/usr/bin/env python3
"""Target-sharded align() with one PROCESS per rank, the mailboxes connected through IPC
handles -- what bench.py's sharded leg does on an 8-GPU node -- here with every rank on GPU
`rank % device_count` (on a one-GPU box all ranks share GPU 0).  Handles travel over a gloo
process group.  Prints one line per rank and a verdict against the unsharded run.

usage: gpu_mailbox_ipc.py [world=2] [points=4000] [mode=cvo|acvo]"""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, n, acvo, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    pkg = ge.load_package()
    capi = pkg.capi
    mode = capi.MODE_ACVO if acvo else capi.MODE_CVO
    xf, ff, xm, fm = pkg.data.synthetic_pair(n, n + n // 7, seed=61, acvo=acvo)
    c = capi.Context(mode=mode, device=dev)
    c.set_fixed(xf, ff)
    c.set_moving(xm, fm)
    lo, hi = capi.shard_range(len(xf), rank, world)
    slo, shi = capi.shard_range(len(xm), rank, world)
    c.set_shard(lo, hi, slo, shi)
    handle, _ = c.mailbox_create(rank, world)
    handles = [None] * world
    dist.all_gather_object(handles, handle)
    c.mailbox_connect(handles=handles)
    dist.barrier()
    res = []
    for _ in range(2):   # twice: the sequence numbers run on from one align() to the next
        st = capi.init_state(c.params)
        it, _ = c.align(st, trace_cap=0)
        res.append((it, bytes(st)))
    fip = c.function_inner_product(0.1) if acvo else 0.0
    dist.barrier()
    c.close()
    out[rank] = (res, fip, np.array(st.transform, np.float32).reshape(4, 4))
    dist.destroy_process_group()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    acvo = len(sys.argv) > 3 and sys.argv[3] == "acvo"
    import numpy as np
    import torch.multiprocessing as mp
    import __graft_entry__ as ge
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.start_processes(worker, args=(world, port, n, acvo, out), nprocs=world, join=True, start_method="spawn")
        res = dict(out)
    pkg = ge.load_package()
    capi = pkg.capi
    mode = capi.MODE_ACVO if acvo else capi.MODE_CVO
    xf, ff, xm, fm = pkg.data.synthetic_pair(n, n + n // 7, seed=61, acvo=acvo)
    ref = capi.Context(mode=mode, device=0)
    ref.set_fixed(xf, ff)
    ref.set_moving(xm, fm)
    st = capi.init_state(ref.params)
    it_ref, _ = ref.align(st, trace_cap=0)
    fip_ref = ref.function_inner_product(0.1) if acvo else 0.0
    T_ref = np.array(st.transform, np.float32).reshape(4, 4)
    ref.close()
    ok = sorted(res) == list(range(world))
    for r in sorted(res):
        runs, fip, T = res[r]
        rot, tra = pkg.data.rel_pose_error(T, T_ref)
        print("rank %d: iterations %s (unsharded %d), rel err rot %.2e trans %.2e, inner product %.9g (unsharded %.9g)"
              % (r, [x[0] for x in runs], it_ref, rot, tra, fip, fip_ref))
        ok = ok and all(x[0] == it_ref for x in runs) and rot <= 1e-6 and tra <= 1e-6
        ok = ok and runs[0][1] == res[0][0][0][1] and runs[1][1] == res[0][0][1][1]   # lock step, bit for bit
        ok = ok and abs(fip - fip_ref) <= 1e-6 * abs(fip_ref)
    print("mailbox ipc world %d: %s" % (world, "OK" if ok else "MISMATCH"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
