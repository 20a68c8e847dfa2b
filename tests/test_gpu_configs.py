"""-m gpu: every BASELINE.json config at its stated size, HIP (through the C-ABI)
against the CPU oracle.

configs[0]  TUM fr1/desk pair decimated to ~3k x 3k        -> tests/test_gpu_parity.py (tum pair)
configs[1]  synthetic 10k x 10k, seed 20190402, cvo + acvo -> here, per-iteration trace equality
configs[2]  fr1/desk streamed (the 5 shipped clouds)       -> tests/test_gpu_paths.py (carry-over)
configs[3]  synthetic 200k x 200k, seed 20191001, sharded  -> here: one iteration of flow / step
            coefficients against the oracle at ell = 0.15 and 0.03, and a whole align()
            two-rank-sharded (device mailboxes) against the unsharded run
configs[4]  8 concurrent 20k x 20k per GPU, seeds 1000 + i -> here, one align_many call

Tolerances as in tests/test_gpu_parity.py: float32 quantities identical, float64 sums
1e-11 relative, transforms 1e-6 (north_star: 1e-4).  ref src/cvo.cpp:361-420.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SUM_RTOL = 1e-11


def _close(a, b, rtol=SUM_RTOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() <= rtol * max(np.abs(b).max(), 1e-300)


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("mode_name", ["cvo", "acvo"])
def test_config1_10k_x_10k_trace_equals_oracle(pkg, po, mode_name):
    """BASELINE configs[1]: the pair bench.py times.  Every iteration's nnz, ell, float32 twist
    and step equal the oracle's; same iteration count; same transform."""
    acvo = mode_name == "acvo"
    mode = pkg.capi.MODE_ACVO if acvo else pkg.capi.MODE_CVO
    xf, ff, xm, fm = pkg.data.synthetic_pair(10000, 10000, seed=pkg.data.SEED_CFG2, acvo=acvo)
    reg = (pkg.Acvo if acvo else pkg.Cvo)(device=0, stream=_stream())
    reg.run_cvo(xf, ff)
    reg.run_cvo(xm, fm, trace_cap=2000)
    p = po.default_params(mode)
    st = po.init_state(p)
    n_or, tr_or = po.align(p, st, xf, ff, xm, fm, search=po.SEARCH_GRID)
    T_or, _, A_or = po.state_matrices(st)
    assert reg.num_iterations == n_or
    assert len(reg.trace) == len(tr_or)
    for a, b in zip(reg.trace, tr_or):
        assert a["nnz"] == b["nnz"] and a["ell"] == b["ell"]
        assert a["nnz_xx"] == b["nnz_xx"] and a["nnz_yy"] == b["nnz_yy"]
        assert a["omega"] == b["omega"] and a["v"] == b["v"] and a["step"] == b["step"]
        assert _close(a["omega_d"], b["omega_d"]) and _close(a["v_d"], b["v_d"])
        assert _close(a["bcde"], b["bcde"], 1e-9)
    rot, tr = pkg.data.rel_pose_error(reg.transform, T_or)
    assert rot <= 1e-6 and tr <= 1e-6
    assert np.array_equal(np.array(reg.state.R), np.array(st.R))
    assert np.array_equal(np.array(reg.state.T), np.array(st.T))
    assert np.allclose(reg.accum_transform, A_or, rtol=0, atol=1e-7)
    reg.close()


@pytest.mark.parametrize("mode_name", ["cvo", "acvo"])
def test_config4_eight_concurrent_20k_x_20k(pkg, po, mode_name):
    """BASELINE configs[4], per GPU: 8 registrations of 20k x 20k (seeds 1000 + i) in flight
    through ONE cvo_hip_align_many call; each equals the oracle's registration of its pair."""
    import torch
    capi = pkg.capi
    acvo = mode_name == "acvo"
    mode = capi.MODE_ACVO if acvo else capi.MODE_CVO
    pairs = [pkg.data.synthetic_pair(20000, 20000, seed=pkg.data.SEED_CFG5_BASE + i, acvo=acvo) for i in range(8)]
    streams = [torch.cuda.Stream() for _ in pairs]
    ctxs = []
    for (xf, ff, xm, fm), s in zip(pairs, streams):
        c = capi.Context(mode=mode, device=0, stream=s.cuda_stream)
        c.set_fixed(xf, ff)
        c.set_moving(xm, fm)
        ctxs.append(c)
    states = [capi.init_state(c.params) for c in ctxs]
    its = capi.align_many(ctxs, states)
    p = po.default_params(mode)
    for i, (xf, ff, xm, fm) in enumerate(pairs):
        st = po.init_state(p)
        n_or, _ = po.align(p, st, xf, ff, xm, fm, search=po.SEARCH_GRID, trace_cap=1)
        assert its[i] == n_or, "pair %d" % i
        assert np.array_equal(np.array(states[i].R), np.array(st.R)), "pair %d" % i
        assert np.array_equal(np.array(states[i].T), np.array(st.T)), "pair %d" % i
        assert states[i].iter == st.iter and states[i].ell == st.ell
        rot, tr = pkg.data.rel_pose_error(np.array(states[i].transform, np.float32).reshape(4, 4),
                                          po.state_matrices(st)[0])
        assert rot <= 1e-6 and tr <= 1e-6
    for c in ctxs:
        c.close()


@pytest.fixture(scope="module")
def cfg3(pkg):
    return pkg.data.synthetic_pair(200000, 200000, seed=pkg.data.SEED_CFG4)


@pytest.mark.parametrize("ell", [0.03, 0.15])
def test_config3_200k_x_200k_one_iteration(pkg, po, cfg3, ell):
    """BASELINE configs[3] at full size (4e10 pairs per sweep): nnz(A) exact, float64 flow sums
    and step coefficients against the oracle, float32 twist identical -- at the widest and the
    narrowest length-scale of the cvo schedule (4.6e8 and 1.4e7 members of A)."""
    xf, ff, xm, fm = cfg3
    c = pkg.capi.Context(mode=pkg.capi.MODE_CVO, device=0, stream=_stream())
    c.set_fixed(xf, ff)
    c.set_moving(xm, fm)
    R, T = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    c.transform_pcd(R, T)
    out = c.flow(ell)
    p = po.default_params(po.MODE_CVO)
    csr = po.se_kernel(p, ell, xf, ff, xm, fm, search=po.SEARCH_GRID)
    om, v, sa, sad2 = po.flow(p, ell, xf, xm, csr)
    assert int(out[8]) == int(csr[0][-1])
    assert _close(out[0:3], om) and _close(out[3:6], v)
    assert _close([out[6]], [sa]) and _close([out[7]], [sad2])
    omega, vv = om.astype(np.float32), v.astype(np.float32)
    assert np.array_equal(out[0:3].astype(np.float32), omega)
    assert np.array_equal(out[3:6].astype(np.float32), vv)
    bcde = c.step_coeffs(omega, vv, ell)
    ref = po.step_coeffs(ell, omega, vv, xf, xm, csr)
    del csr
    assert _close(bcde, ref, 1e-10)
    assert pkg.capi.pick_step(bcde) == po.pick_step(ref)
    c.close()


def test_config3_200k_x_200k_two_ranks_equal_one(pkg):
    """BASELINE configs[3]: a whole align() with the target rows split over two ranks -- two
    contexts on this GPU, their 13 + 4 float64 partial sums exchanged through device-memory
    mailboxes inside the launch chain (the xGMI peer-store all-reduce of SURVEY 8e, here with
    both mailboxes on one device) -- against the unsharded run: same iteration count, both
    ranks bit-identical, transform within 1e-6.  In a process of its own (tools/gpu_ranks_threads.py): two ranks that
    spin for each other inside kernels need hardware queues of their own, which only the first streams of a process are sure of."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_ranks_threads.py"), "cvo", "200000", "200000", "2", "classic",
                        str(pkg.data.SEED_CFG4)], capture_output=True, text=True, timeout=900, env=dict(os.environ, GPU_MAX_HW_QUEUES="8"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "ranks on one gpu, world 2, cvo 200000 x 200000: OK" in r.stdout


def test_headline_shape_64_distinct_10k_pairs_through_the_engines(pkg, po):
    """The shape bench.py's `value` is quoted on: 64 DISTINCT 10k x 10k pairs in ONE align_many call
    (three engines of 21-22 slots, candidate lists, captured batches).  Every registration equals the
    same pair registered on its own (cvo_hip_align: head mode, asynchronous builds) bit for bit, and
    four of them equal the oracle's (ref src/cvo.cpp:361-420)."""
    import torch
    capi = pkg.capi
    count = 64
    ctxs, streams, pairs = [], [], []
    for b in range(count):
        seed = pkg.data.SEED_CFG2 if b == 0 else pkg.data.SEED_CFG5_BASE + b
        pr = pkg.data.synthetic_pair(10000, 10000, seed=seed)
        s = torch.cuda.Stream()
        c = capi.Context(mode=capi.MODE_CVO, device=0, stream=s.cuda_stream, graph_capture=True)
        c.set_fixed(pr[0], pr[1])
        c.set_moving(pr[2], pr[3])
        ctxs.append(c); streams.append(s); pairs.append(pr)
    for rep in range(2):   # (the second call re-uses the engines' captured batches and tables)
        states = [capi.init_state(c.params) for c in ctxs]
        its = capi.align_many(ctxs, states)
    assert len(set(its)) > 8   # distinct pairs: a spread of iteration counts, slots refilled as they fall free
    # (round 6: the call's last registrations leave the engines for resident runs of their own -- csrc/cvo_engine.cpp "the call's tail" --;
    # the comparison below covers them: same state as cvo_hip_align, bit for bit)
    assert sum(c.get_option("tail_handovers") for c in ctxs) >= 1
    for b, c in enumerate(ctxs):
        st = capi.init_state(c.params)
        n_l, _ = c.align(st, trace_cap=0)
        assert n_l == its[b], (b, n_l, its[b])
        assert bytes(st) == bytes(states[b]), b
    p = po.default_params(po.MODE_CVO)
    for b in (0, 1, 17, 63):
        xf, ff, xm, fm = pairs[b]
        so = po.init_state(p)
        n_or, _ = po.align(p, so, xf, ff, xm, fm, search=po.SEARCH_GRID, trace_cap=1)
        assert n_or == its[b], (b, n_or, its[b])
        assert np.array_equal(np.array(states[b].R), np.array(so.R)), b
        assert np.array_equal(np.array(states[b].T), np.array(so.T)), b
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("launcher", ["torchrun", "bare"])
def test_bench_multi_rank_rehearsal_on_one_gpu(launcher):
    """The driver's 8-GPU command, rehearsed with two ranks on device 0 (CVO_BENCH_RANKS_ON_DEVICE0=1:
    gloo for torch's collectives): the weak-scaling leg, the all_gather of the IPC handles, the mailbox
    leg of the target-sharded mode (ref src/cvo.cpp:201-204,283-288 across ranks), its watchdog and
    the assembly of the JSON line all run before the driver runs them for the first time on 8 GPUs.
    Both launch conventions: under torch.distributed.run, and a bare `python bench.py --gpus 2`, which
    re-executes itself under the launcher -- the same line either way."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CVO_BENCH_RANKS_ON_DEVICE0="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    bench_args = [os.path.join(root, "bench.py"),
                  "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--points", "3000",
                  "--sharded-points", "20000", "--sharded-steps", "1", "--sharded-timeout", "120",
                  "--config4-points", "4000", "--config4-count", "4"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", "29577"] + bench_args
    else:
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable] + bench_args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip().splitlines()[-1] == lines[0]   # ONE JSON line, last
    assert len(lines[0]) < 6000   # (inside the driver's 8 KB tail whole)
    head = json.loads(lines[0])
    assert head["n_gpus"] == 2 and head["scaling"] == "weak" and head["value"] > 0
    assert head["ranks_on_device0"] is True
    assert head["config"]["config4_registrations_per_s"] > 0
    assert head["config"]["one_registration_at_a_time"]["registrations_per_s"] > 0
    assert "error" not in head["sharded_allreduce"], head["sharded_allreduce"]
    assert "leg_errors" not in head, head["leg_errors"]
    # ... and the full object beside it
    with open(os.path.join(root, head["detail"])) as fh:
        out = json.load(fh)
    assert out["value"] == pytest.approx(head["value"], rel=1e-5)
    # BASELINE configs[4] under N ranks (every rank its own registrations, max-over-ranks clock) and the literal one-pair
    # figure of configs[1], both also inside `config` (the part of the line the driver keeps whole)
    c4 = out["config4"]
    assert "error" not in c4, c4
    assert c4["n_gpus"] == 2 and c4["registrations_per_s"] > 0 and c4["iterations_per_registration"] > 10
    assert out["config"]["config4_registrations_per_s"] == c4["registrations_per_s"]
    assert head["config"]["config4_registrations_per_s"] == pytest.approx(c4["registrations_per_s"], rel=1e-5)
    one = out["config"]["one_registration_at_a_time"]
    assert one["registrations_per_s"] > 0 and one["ms_per_iteration"] > 0 and one["iterations"] > 10
    sh = out["sharded_allreduce"]
    assert "error" not in sh, sh
    assert sh["exchange"] == "mailbox" and sh["registrations_per_s"] > 0, sh
    assert sh["iterations"] > 0
