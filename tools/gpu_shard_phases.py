#!/usr/bin/env python3
"""The per-GPU problem of BASELINE configs[3] on 8 GPUs (rank 0's rows of an n x n pair against the whole source, a world of one
through the mailbox path), ms per iteration by length scale.  CVO_HIP_NO_MERGE=1: the five-launch scheme of rounds 2-4
(post-flow launch with the exchange); default: four launches, the flow-side exchange inside k_step_twist.
usage: gpu_shard_phases.py [n] [world]"""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
import bench

pkg = ge.load_package(); capi = pkg.capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=pkg.data.SEED_CFG4)
c = capi.Context(mode=capi.MODE_CVO, device=0, graph_capture=True)
c.set_fixed(xf, ff); c.set_moving(xm, fm)
lo, hi = capi.shard_range(n, 0, world); slo, shi = capi.shard_range(n, 0, world)
c.set_shard(lo, hi, slo, shi)
c.mailbox_create(0, 1); c.mailbox_connect(ptrs=[None])
r = bench.per_length_scale_ms(c, capi, torch)
print("shard of %d, %dk x %dk, %s: " % (world, n // 1000, n // 1000, "five launches (CVO_HIP_NO_MERGE)" if os.environ.get("CVO_HIP_NO_MERGE") else "four launches") +
      "  ".join("%s %.1f us" % (k, v["ms_per_iteration"] * 1e3) for k, v in r.items()))
c.close()
