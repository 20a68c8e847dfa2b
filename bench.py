#!/usr/bin/env python3
"""bench.py -- frame-pair registrations/s of the CVO inner loop on MI355X.

A "step" is one pass of the hot path over one BATCH of synthetic input: `--batch`
(default 32) independent frame pairs of BASELINE.json configs[1] -- the seeded
synthetic 10k x 10k RGB-D cloud pair -- each run through a full align()
(ref src/cvo.cpp:361-420: ~50 gradient-flow iterations, each = transform +
all-pairs neighbour filter + flow pass + step-size pass) from the reference
object's initial state, all clouds already resident in HBM, all registrations
of the batch in flight at once (cvo_hip_align_many: groups of up to 16
registrations share every kernel launch, blockIdx.z = registration; the groups
run on their own streams and fill each other's bubbles).  `value` = registrations completed per second; the
single-registration latency (batch of one) is measured in the same run and
reported as `single_stream`.  One process per GPU; for N > 1 every rank runs its
own batches (independent frame pairs: weak scaling, no data-path collective),
and -- as a separately reported leg -- all ranks also run the target-sharded
mode whose twist / step-coefficient partial sums are all-reduced with RCCL
(BASELINE.json configs[3] scaled to fit the time budget).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

FLOP_PER_PAIR = 8.0            # 3 sub + 3 mul + 2 add per pair test
BYTES_PER_POINT = 32.0         # xyz 12 B + 5 features 20 B
PEAK_F32_TFLOPS = 157.3        # MI355X_MICROARCH.md: FP32 vector = FP32 MFMA peak
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=10000, help="N = M of the synthetic pair")
    ap.add_argument("--batch", type=int, default=32, help="frame pairs in flight per step")
    ap.add_argument("--mode", default="cvo", choices=["cvo", "acvo"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="budget of the cpu_baseline leg (rank 0, N=1 only)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-frontend", action="store_true", help="skip the RGB-D front end side leg")
    ap.add_argument("--sharded-points", type=int, default=40000)
    ap.add_argument("--sharded-steps", type=int, default=2)
    ap.add_argument("--sharded-timeout", type=int, default=240, help="watchdog of the sharded leg, seconds")
    ap.add_argument("--force-sharded-leg", action="store_true",
                    help="run the RCCL-sharded leg even on one GPU (world size 1): exercises the "
                         "multi-rank code path where only one GPU is available")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    torch.cuda.set_device(local_rank)
    if world == 1 and args.force_sharded_leg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    pkg = ge.load_package()
    capi = pkg.capi
    acvo = args.mode == "acvo"
    mode = capi.MODE_ACVO if acvo else capi.MODE_CVO
    n = m = args.points
    # every rank registers its own frame pairs: `batch` contexts, one stream each
    # (all pairs are the configs[1] pair itself: identical work per registration)
    xf, ff, xm, fm = pkg.data.synthetic_pair(n, m, seed=pkg.data.SEED_CFG2, acvo=acvo)
    B = max(1, args.batch)
    streams = [torch.cuda.Stream() for _ in range(B)]
    ctxs = []
    for b in range(B):
        c = capi.Context(mode=mode, device=local_rank, stream=streams[b].cuda_stream)
        c.set_fixed(xf, ff)
        c.set_moving(xm, fm)
        ctxs.append(c)
    ctx = ctxs[0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step(cs):
        states = [capi.init_state(c.params) for c in cs]
        its = capi.align_many(cs, states)
        return its, states

    for _ in range(args.warmup):
        one_step(ctxs)
    barrier()
    t0 = time.perf_counter()
    iters = 0
    last_state = None
    for _ in range(args.steps):
        its, states = one_step(ctxs)
        iters += sum(its)
        last_state = states[0]
    barrier()
    elapsed = time.perf_counter() - t0

    # the same pair, one registration at a time (latency view), same run
    for _ in range(2):
        one_step(ctxs[:1])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    it1 = 0
    n1 = max(5, args.steps)
    for _ in range(n1):
        its, _ = one_step(ctxs[:1])
        it1 += its[0]
    torch.cuda.synchronize()
    el1 = time.perf_counter() - t1
    single = {"registrations_per_s": n1 / el1, "ms_per_registration": el1 * 1e3 / n1,
              "ms_per_iteration": el1 * 1e3 / max(it1, 1)}

    # roofline leg: HIP events on context 0's own stream around every k_filter launch,
    # one registration in flight so that a launch has the device to itself (a kernel
    # duration measured while B streams share the CUs is not a roofline input)
    ctx.set_profiling(True)
    ctx.get_profile(reset=True)
    for _ in range(max(3, args.steps // 4)):
        one_step(ctxs[:1])
    torch.cuda.synchronize()
    prof = ctx.get_profile(reset=True)
    ctx.set_profiling(False)

    t_max = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    it_sum = torch.tensor([float(iters)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(it_sum, op=dist.ReduceOp.SUM)
    elapsed = float(t_max.item())
    total_regs = args.steps * world * B
    value = total_regs / elapsed

    # parity sanity inside the bench: the registration recovers the synthetic motion
    T_est = np.array(last_state.transform, np.float64).reshape(4, 4)
    rot_err, tr_err = pkg.data.rel_pose_error(np.linalg.inv(T_est), np.linalg.inv(pkg.data.gt_motion()))

    out = None
    if rank == 0:
        # the dominant kernel: k_filter on the (fixed x moving) pair set, one launch
        # per executed iteration, bracketed by HIP events on the context's stream
        launches = prof["flow_launches"]
        sweep_ms = prof["flow_ms"] / max(launches, 1)
        pairs = prof["flow_pairs"] / max(launches, 1)
        achieved = FLOP_PER_PAIR * pairs / (sweep_ms * 1e-3) / 1e12 if sweep_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as fh:
                    traffic = json.load(fh).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # which kernels the time of the batched run goes to, from the committed rocprofv3 summary of
        # this same command (profiles/, refreshed by tools/gpu_profile.sh): informational
        shares = None
        spath = os.path.join(ROOT, "profiles", "r01_kernel_stats_batch32.csv")
        if os.path.exists(spath):
            try:
                import csv
                rows = [r for r in csv.DictReader(open(spath)) if "cvo_dev::" in r["Name"]]
                tot = sum(float(r["TotalDurationNs"]) for r in rows)
                top = sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:5]
                shares = {r["Name"].replace("void ", "").replace("cvo_dev::", "").split("(")[0]:
                          round(float(r["TotalDurationNs"]) / tot, 3) for r in top}
            except Exception:
                shares = None
        # the same kernel's duration in the committed rocprofv3 kernel trace of this command
        # (executed launches only): HIP events read 2-5 us more (the event pair's own handling)
        rocprof_us = None
        lpath = os.path.join(ROOT, "profiles", "r01_kernel_live.json")
        if os.path.exists(lpath):
            try:
                with open(lpath) as fh:
                    rocprof_us = json.load(fh).get("k_filter", {}).get("live_avg_us")
            except Exception:
                rocprof_us = None
        algo_bytes = BYTES_PER_POINT * (n + m)
        iters_per_reg = float(it_sum.item()) / total_regs
        # two all-pairs sweeps (flow, step size) per iteration, 8 flop per
        # pair test.  The path culls tile pairs, re-uses its neighbour lists across
        # iterations and evaluates the step-size sweep on the members of A only, so this
        # is an EQUIVALENT rate (work the reference's dense formulation would do / time).
        sweep_flop_per_iter = 2.0 * FLOP_PER_PAIR * float(n) * m
        equiv_batched = sweep_flop_per_iter * float(it_sum.item()) / elapsed / 1e12
        equiv_single = sweep_flop_per_iter / (single["ms_per_iteration"] * 1e-3) / 1e12
        out = {
            "metric": "frame-pair registrations/sec",
            "value": value,
            "unit": "registrations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "synthetic %dk x %dk RGB-D cloud pair (xyz + 5-dim colour), %s align() "
                            "to convergence; a step = a batch of %d such registrations in flight"
                            % (n // 1000, m // 1000, args.mode, B),
                "points_fixed": n, "points_moving": m, "mode": args.mode,
                "pairs_per_sweep": float(n) * m,
                "batch": B,
                "parallelism": "%d independent registrations in flight per GPU, fused in groups of <= 16 into "
                               "shared kernel launches (blockIdx.z = registration), one stream per group" % B,
            },
            "iterations_per_registration": iters_per_reg,
            "equivalent_sweep_rate": {
                "definition": "2 sweeps x 8 flop x N x M per iteration / time (SURVEY 8d b); exceeds the "
                              "f32 peak because most pair tests are proven unnecessary, not executed",
                "batched_TFLOPs": equiv_batched, "single_stream_TFLOPs": equiv_single,
                "peak_TFLOPs": PEAK_F32_TFLOPS},
            "ms_per_iteration": elapsed * 1e3 * world / max(float(it_sum.item()), 1.0),
            "single_stream": single,
            "gt_motion_rel_err": {"rot": rot_err, "trans": tr_err},
            "kernel_time_shares_batched": shares,
            "roofline": {
                "kernel": "cvo_dev::k_filter (all target x source pair tests, v_mfma_f32_16x16x4_f32)",
                "bound": "mfma",
                "pipe": "f32 MFMA issue (f32 MFMA peak == f32 vector peak on gfx950)",
                "achieved": achieved,
                "peak": PEAK_F32_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_TFLOPS,
                "flop_per_launch": FLOP_PER_PAIR * pairs,
                "avg_launch_us": sweep_ms * 1e3,
                "rocprofv3_avg_launch_us": rocprof_us,
                "frac_at_rocprofv3_duration": (FLOP_PER_PAIR * pairs / (rocprof_us * 1e-6) / 1e12 / PEAK_F32_TFLOPS)
                if rocprof_us else None,
                "launches": launches,
                "launches_note": "k_filter does work only in the iterations that rebuild the tile list "
                                 "(%.1f of %.1f iterations per registration here); the other launches "
                                 "return at once and are not counted" % (
                                     launches / float(max(3, args.steps // 4)), iters_per_reg),
                "measured": "HIP events on the launching stream, every k_filter launch of %d "
                            "single-stream registrations of the same run" % max(3, args.steps // 4),
                "traffic": traffic,
            },
            "roofline_hbm": {
                "bound": "hbm",
                "achieved": algo_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0,
                "peak": PEAK_HBM_GBS,
                "unit": "GB/s",
                "frac": (algo_bytes / (sweep_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if sweep_ms > 0 else 0.0,
                "algorithmic_bytes_per_launch": algo_bytes,
            },
        }
        if world == 1 and not args.no_frontend:
            for c in ctxs:   # (dozens of idle streams slow every other stream's submissions down)
                c.close()
            ctxs = []
            try:
                out["frontend"] = frontend_leg(args, pkg)
            except Exception as e:   # the headline line must survive a side leg
                out["frontend"] = {"error": repr(e)}
        # (last: the OpenMP team of the CPU leg keeps spinning for a while after its last
        # parallel region and would slow the host side of everything timed after it)
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args, pkg, xf, ff, xm, fm, acvo)
    # The target-sharded leg (RCCL all-reduce per iteration) runs last and under a watchdog:
    # the headline line above it must not depend on it, not even if a collective hangs.
    if (world > 1 or args.force_sharded_leg) and args.sharded_steps > 0:
        import threading

        def give_up():
            if rank == 0:
                out["sharded_allreduce"] = {"error": "timed out after %d s" % args.sharded_timeout}
                print(json.dumps(out), flush=True)
            os._exit(0)

        if world > 1:
            dist.barrier()
        dog = threading.Timer(args.sharded_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            sharded = sharded_leg(args, pkg, dist, torch, rank, world, local_rank, barrier)
        except Exception as exc:
            sharded = {"error": repr(exc)}
        dog.cancel()
        if rank == 0:
            out["sharded_allreduce"] = sharded
    if rank == 0:
        print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if world > 1 or args.force_sharded_leg:
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
    return out


def frontend_leg(args, pkg, frames=100):
    """Side leg (not part of `value`): the RGB-D front end on synthetic VGA
    frames, host images in / host cloud out (PCIe inclusive), and its CPU restatement
    (oracle, one thread) on the same frames."""
    imgs = [pkg.data.synthetic_rgbd_frame(seed=100 + k, texture=1.0) for k in range(4)]
    gen = pkg.frontend.PcdGenerator(640, 480)
    for bgr, dep in imgs:
        gen.create_pointcloud(bgr, dep)
    t0 = time.perf_counter()
    for k in range(frames):
        xyz, _ = gen.create_pointcloud(*imgs[k % 4])
    dt = (time.perf_counter() - t0) / frames
    out = {"frames_per_s": 1.0 / dt, "ms_per_frame": dt * 1e3, "image": "640x480 synthetic, texture 1.0",
           "points": int(len(xyz)), "includes": "staging + PCIe in, kernels, cloud out"}
    # the reference's driver loop end to end (ref src/cvo_main.cpp:36-66) on a synthetic
    # sequence: image pair -> front end -> run_cvo -> pose
    seq = [("%d" % k,) + pkg.data.synthetic_rgbd_frame(seed=77, texture=1.0, motion=(1.5 * k, 0.7 * k))
           for k in range(12)]
    stream = {}
    for name, cls in (("cvo", pkg.Cvo), ("acvo", pkg.Acvo)):
        passes = []
        for _ in range(3):   # three passes on fresh objects, the median one is reported
            reg = cls()
            pkg.frontend.run_frames(reg, seq[:3], 1, generator=gen)   # warm-up
            reg.close()
            reg = cls()
            t0 = time.perf_counter()
            pkg.frontend.run_frames(reg, seq * 3, 1, generator=gen)
            passes.append((time.perf_counter() - t0) / (3 * len(seq)))
            reg.close()
        dt = sorted(passes)[1]
        stream[name] = {"frames_per_s": 1.0 / dt, "ms_per_frame": dt * 1e3,
                        "passes_ms_per_frame": [round(v * 1e3, 3) for v in passes]}
    out["stream"] = stream
    out["stream_note"] = "36 synthetic VGA frames (12, three times over), ~3k points each, decoded images in host memory, one frame at a time; median of 3 passes"
    gen.close()
    if not args.no_cpu:
        from oracle import pyoracle_fe as fo
        t0 = time.perf_counter()
        for k in range(4):
            fo.create_pointcloud(*imgs[k])
        out["cpu_ms_per_frame"] = (time.perf_counter() - t0) / 4 * 1e3
        out["cpu_kind"] = "port, 1 thread"
    return out


def sharded_leg(args, pkg, dist, torch, rank, world, local_rank, barrier):
    """Target rows sharded over the ranks; the 13 + 4 float64 partial sums are
    all-reduced with RCCL inside the C-ABI twice per iteration."""
    capi = pkg.capi
    acvo = args.mode == "acvo"
    n = m = args.sharded_points
    xf, ff, xm, fm = pkg.data.synthetic_pair(n, m, seed=pkg.data.SEED_CFG4, acvo=acvo)
    stream = torch.cuda.current_stream().cuda_stream
    ctx = capi.Context(mode=capi.MODE_ACVO if acvo else capi.MODE_CVO, device=local_rank,
                       stream=stream)
    ctx.set_fixed(xf, ff)
    ctx.set_moving(xm, fm)
    lo, hi = capi.shard_range(n, rank, world)
    slo, shi = capi.shard_range(m, rank, world)
    ctx.set_shard(lo, hi, slo, shi)
    uid = [capi.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    ctx.comm_init(uid[0], rank, world)
    st = capi.init_state(ctx.params)
    ctx.align(st, trace_cap=0)   # warm-up (also sets up the RCCL channels)
    barrier()
    t0 = time.perf_counter()
    iters = 0
    for _ in range(args.sharded_steps):
        st = capi.init_state(ctx.params)
        n_it, _ = ctx.align(st, trace_cap=0)
        iters += n_it
    barrier()
    el = time.perf_counter() - t0
    t = torch.tensor([el], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    ctx.close()
    return {"workload": "synthetic %dk x %dk, target rows sharded %d ways, RCCL all-reduce of "
                        "13+4 float64 per iteration" % (n // 1000, m // 1000, world),
            "scaling": "strong", "registrations_per_s": args.sharded_steps / el,
            "ms_per_iteration": el * 1e3 / max(iters, 1), "iterations": iters / args.sharded_steps}


def cpu_baseline(args, pkg, xf, ff, xm, fm, acvo):
    """The oracle (kind "port": the reference cannot be built here) timed on the
    host cores of this box on a bounded sample of the same workload."""
    from oracle import pyoracle as po
    p = po.default_params(po.MODE_ACVO if acvo else po.MODE_CVO)
    # the restatement's OpenMP regions are short: more threads than it can feed
    # make it slower, so calibrate the thread count on one registration each
    po.set_threads(0)
    ncpu = po.get_threads()
    best, cores = None, 1
    for nt in sorted({1, 8, 16, 32, 64, ncpu}):
        if nt > ncpu:
            continue
        po.set_threads(nt)
        t0 = time.perf_counter()
        st = po.init_state(p)
        po.align(p, st, xf, ff, xm, fm, search=po.SEARCH_GRID, trace_cap=1)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, nt
        if dt > 4 * best:
            break
    po.set_threads(cores)
    done, iters = 0, 0
    t0 = time.perf_counter()
    while True:
        st = po.init_state(p)
        n_it, _ = po.align(p, st, xf, ff, xm, fm, search=po.SEARCH_GRID, trace_cap=1)
        done += 1
        iters += n_it
        el = time.perf_counter() - t0
        if el >= args.cpu_seconds or done >= 50:
            break
    return {"value": done / el, "unit": "registrations/s", "cores": cores, "kind": "port",
            "ms_per_iteration": el * 1e3 / iters,
            "sample": "%d full registration(s) of the same %dk x %dk pair (%d iterations), "
                      "uniform-grid radius search + CSR Gram matrix as the reference, OpenMP on %d "
                      "threads, %.1f s" % (done, xf.shape[0] // 1000, xm.shape[0] // 1000, iters,
                                          cores, el)}


if __name__ == "__main__":
    main()
