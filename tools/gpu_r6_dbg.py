import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, __graft_entry__ as ge
pkg = ge.load_package(); capi = pkg.capi
if os.environ.get("CVO_LIB"): capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), os.environ["CVO_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
acvo = bool(os.environ.get("ACVO"))
xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=pkg.data.SEED_CFG2, acvo=acvo)
c = capi.Context(mode=capi.MODE_ACVO if acvo else capi.MODE_CVO, device=0)
c.set_option("run_timeout_ms", 50)
if os.environ.get("NO_RESTART"): c.set_option("run_restart", 0)
c.set_fixed(xf, ff); c.set_moving(xm, fm)
for rep in range(4):
    st = capi.init_state(c.params); t = time.perf_counter()
    try:
        it, _ = c.align(st, trace_cap=0)
    except Exception as e:
        it = -1; print("   ", e)
    torch.cuda.synchronize()
    print("rep", rep, "iters", it, "ms %.2f" % ((time.perf_counter() - t) * 1e3), "runs", c.run_stats(), "timeouts", c.get_option("run_timeouts"),
          "side launched", c.get_option("side_builds_launched"), "backoff", c.get_option("no_run_backoff"), "dbg", c.run_clocks()[:12])
c.close()
