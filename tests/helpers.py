"""Shared drivers of the -m gpu tests: a target-sharded align() with a REAL second rank on
a one-GPU box (two contexts, two host threads)."""
import ctypes
import threading

import numpy as np


def align_two_ranks(pkg, mode, xf, ff, xm, fm, exchange="mailbox", timeout=300, world=2):
    """Runs `world` contexts on GPU 0, context r holding rows shard_range(n, r, world) of the
    fixed cloud (and of the moving cloud for the acvo Ayy rows), each from its own host thread.

    exchange = "mailbox": cvo_hip_mailbox_create / _connect -- the partial sums travel through
               device-memory mailboxes inside the launch chain (the peer-store all-reduce that
               runs over xGMI between GPUs; here every mailbox lives on the one device);
    exchange = "hook":    cvo_hip_set_allreduce with a host-memory sum in rank order.

    Returns {rank: (iterations, bytes(state), transform 4x4, all-reduce calls)}."""
    capi = pkg.capi
    import torch
    n, m = len(xf), len(xm)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    barrier = threading.Barrier(world)
    box = [None] * world
    boxes = [None] * world
    out = {}
    calls = [0] * world
    errors = []

    def rank(r):
        try:
            # (a stream of the library's own, created here and now: the ranks' streams are made back to back and land on hardware
            # queues of their own.  Streams from torch's pool -- 32 of them, handed out in turn to whoever asks in the process --
            # put two ranks on ONE hardware queue on some boxes: a rank that spins for its peer inside a kernel then keeps that
            # peer's kernels from ever starting, and both time out.  Ranks that share a GPU exist in tests only.)
            import os
            s = torch.cuda.Stream() if os.environ.get("RANK_TORCH_STREAM") else None
            c = capi.Context(mode=mode, device=0, stream=s.cuda_stream if s is not None else None)
            c.set_fixed(xf, ff)
            c.set_moving(xm, fm)
            lo, hi = capi.shard_range(n, r, world)
            slo, shi = capi.shard_range(m, r, world)
            c.set_shard(lo, hi, slo, shi)
            if exchange == "mailbox":
                boxes[r] = c.mailbox_create(r, world)[1]
                barrier.wait()
                c.mailbox_connect(ptrs=list(boxes))
                barrier.wait()
            else:
                def hook(ptr, count, stream):
                    calls[r] += 1
                    hip.hipStreamSynchronize(stream)
                    mine = np.zeros(count, np.float64)
                    assert hip.hipMemcpy(mine.ctypes.data, ptr, count * 8, 2) == 0      # device -> host
                    box[r] = mine
                    barrier.wait()
                    total = box[0].copy()
                    for q in range(1, world):
                        total = total + box[q]                                          # rank order
                    barrier.wait()
                    assert hip.hipMemcpy(ptr, total.ctypes.data, count * 8, 1) == 0     # host -> device
                c.set_allreduce(hook)
            st = capi.init_state(c.params)
            it, _ = c.align(st, trace_cap=0)
            out[r] = (it, bytes(st), np.array(st.transform, np.float32).reshape(4, 4), calls[r])
            barrier.wait()   # nobody frees a mailbox a peer may still be writing to
            c.close()
        except Exception as e:   # pragma: no cover
            errors.append((r, repr(e)))
            barrier.abort()

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=timeout)
    assert not errors, errors
    assert sorted(out) == list(range(world)), "a rank did not finish"
    return out


def align_two_ranks_mailbox(pkg, mode, xf, ff, xm, fm, **kw):
    return align_two_ranks(pkg, mode, xf, ff, xm, fm, exchange="mailbox", **kw)
