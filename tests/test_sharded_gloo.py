"""CPU, world_size 2, gloo: what a CPU box can legitimately cover of the target-sharded mode.

Covered here: the sharding CONTRACT -- rows split with the product's own cvo_hip_shard_range
(a host function of libcvo_hip.so, no GPU needed), 13 + 4 float64 partial sums summed over the
ranks twice per iteration, every rank running the O(1) maths on the reduced sums --
gives the single-rank iteration count, the global nnz in every iteration, a transform within
1e-6, and ranks that stay bit-identical.  The kernels under the contract are the ORACLE's (the
HIP kernels need a GPU): this test says nothing about the HIP library's own exchange.

Covered on the GPU (-m gpu): the HIP library's row shards add up (test_row_shards_add_up_to_the_whole),
its in-kernel mailbox all-reduce with 2 and 4 real ranks on one device and with one process per
rank over IPC handles (test_ranks_on_one_gpu_through_device_mailboxes,
test_ranks_in_separate_processes_through_ipc_mailboxes), the RCCL and hook paths
(test_rccl_world_size_one_and_user_hook, test_two_ranks_on_one_gpu_through_the_allreduce_hook), and
BASELINE configs[3] at full size (tests/test_gpu_configs.py).  Only an 8-GPU node can show the
peer stores crossing xGMI."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, mode, n, m, out):
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    from oracle import pyoracle as po
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = ge.load_package()
    po.set_threads(2)
    xf, ff, xm, fm = pkg.data.synthetic_pair(n, m, seed=13, acvo=(mode == 1))
    p = po.default_params(mode)
    s = po.init_state(p)
    lo, hi = pkg.capi.shard_range(n, rank, world)
    slo, shi = pkg.capi.shard_range(m, rank, world)
    calls = [0]

    def allreduce(arr):
        t = torch.from_numpy(arr)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        calls[0] += 1

    n_it, tr = po.align(p, s, xf, ff, xm, fm, search=po.SEARCH_GRID,
                        shard=(lo, hi, slo, shi), allreduce=allreduce)
    T = po.state_matrices(s)[0]
    out[rank] = (n_it, T.copy(), calls[0], [t["nnz"] for t in tr])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,n,m", [(0, 700, 650), (1, 500, 620)])
def test_two_rank_sharded_equals_single_rank(mode, n, m):
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    from oracle import pyoracle as po
    pkg = ge.load_package()
    xf, ff, xm, fm = pkg.data.synthetic_pair(n, m, seed=13, acvo=(mode == 1))
    p = po.default_params(mode)
    s = po.init_state(p)
    n_ref, tr_ref = po.align(p, s, xf, ff, xm, fm, search=po.SEARCH_GRID)
    T_ref = po.state_matrices(s)[0]

    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, mode, n, m, out), nprocs=world, join=True)
        res = dict(out)
    assert set(res) == {0, 1}
    for r in range(world):
        n_it, T, calls, nnz = res[r]
        assert n_it == n_ref                       # same iteration count on every rank
        assert calls == 2 * n_ref                  # two all-reduces per iteration
        assert nnz == [t["nnz"] for t in tr_ref]   # the reduced nnz is the global one
        rot, tra = pkg.data.rel_pose_error(T, T_ref)
        assert rot <= 1e-6 and tra <= 1e-6
    assert np.array_equal(res[0][1], res[1][1])    # ranks stay in lock step bit for bit
