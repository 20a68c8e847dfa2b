#!/usr/bin/env python3
"""Round-3 profile summaries: from the raw rocprofv3 outputs of tools/gpu_profile_r3.sh (gpurun_out/<tag>/) to
the small files under profiles/ (or, with --keep-in-out, into gpurun_out/<tag>/summary/ on the GPU box, from
where they are copied).  MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE counts half the
bytes of wide reads on gfx950 (doubled here)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(src, "summary") if "--keep-in-out" in sys.argv else os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
CLOCK_GHZ = 2.4      # MI355X peak engine clock (MI355X_MICROARCH.md)
SIMDS = 256 * 4


def one(pattern):
    g = glob.glob(os.path.join(src, pattern))
    return g[0] if g else None


def short(name):
    return name.split("(")[0].replace("cvo_dev::", "").replace("void ", "")


for pat, name in (("bench.json", "%s_bench.json"), ("stats/*kernel_stats.csv", "%s_kernel_stats.csv"),
                  ("stats_batch/*kernel_stats.csv", "%s_kernel_stats_batch.csv"),
                  ("stats_roofline/*kernel_stats.csv", "%s_kernel_stats_roofline.csv"),
                  ("stats_fe/*kernel_stats.csv", "%s_kernel_stats_frontend.csv"),
                  ("roofline_plain.json", "%s_roofline_only.json"), ("bench_line.json", "%s_bench_line.json"),
                  ("stats_acvo/*kernel_stats.csv", "%s_kernel_stats_acvo.csv")):
    f = one(pat)
    if f:
        shutil.copy(f, os.path.join(dst, name % tag))


def per_dispatch(pattern, counter=None):
    """{kernel: [value per dispatch, in dispatch order]}"""
    f = one(pattern)
    out = collections.defaultdict(list)
    if not f:
        return out
    rows = list(csv.DictReader(open(f)))
    key = "Dispatch_Id" if rows and "Dispatch_Id" in rows[0] else None
    if key:
        rows.sort(key=lambda r: int(r[key]))
    for r in rows:
        if counter is None:
            out[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        elif r["Counter_Name"] == counter:
            out[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return out


# ---- traffic of the list passes of ONE engine of 22 pairs, the same launches in every pass, per length scale
rl = None
try:
    rl = json.loads(open(os.path.join(src, "roofline_plain.json")).read().strip().splitlines()[-1])
except Exception:
    pass
dur = per_dispatch("stats_roofline/*kernel_trace.csv")
fetch = per_dispatch("pmc_fetch_r/*counter_collection.csv", "FETCH_SIZE")
write = per_dispatch("pmc_write_r/*counter_collection.csv", "WRITE_SIZE")
phases = {}
CALLS = 6   # bench.py --roofline-only: one warm-up call + 5
PH = (("ell_0.15", 0, 4), ("ell_0.10", 4, 11), ("ell_0.06", 11, 21), ("ell_0.03", 21, 10 ** 9))


def by_iteration(v):
    """A pass's dispatches of one kernel -> {iteration: [values]}.  Every registration of a call starts at
    iteration 0 together (one engine, all 22 inserted at once), so dispatch i of a call IS iteration i; a pass
    may end its calls with a different number of launches that return at once (the host notices the last
    `done` a batch earlier or later), so every pass is cut into calls by its OWN count."""
    if not v or len(v) % CALLS:
        return None, None
    per = len(v) // CALLS
    it = collections.defaultdict(list)
    for c in range(CALLS):
        for i in range(per):
            it[i].append(v[c * per + i])
    return it, per


for k in ("kt_process<0, 0>", "kt_process<1, 0>", "kt_filter", "kt_post_step", "kt_post_flow", "kt_step_twist"):
    d, f, w = dur.get(k, []), fetch.get(k, []), write.get(k, [])
    if not d or not f or not w:
        continue
    mean = lambda x: sum(x) / max(len(x), 1)
    us, fb, wb = mean(d), 2.0 * mean(f) * 1024.0, mean(w) * 1024.0
    ph = {"all": {"launches_trace_fetch_write_pass": [len(d), len(f), len(w)], "avg_us": us, "fetch_bytes_per_launch": fb,
                  "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
                  "what": "every launch of the command, each pass averaged over its own launches (the launches that return at "
                          "once -- queued past the last convergence -- are in all three averages)"}}
    di, dper = by_iteration(d)
    fi, fper = by_iteration(f)
    wi, wper = by_iteration(w)
    if di and fi and wi:
        # (ref src/cvo.cpp:408-410, applied at the END of iteration k: ell = 0.15 in iterations 0-3, 0.10 in 4-10, 0.06 in 11-20, then 0.03;
        # round 3 cut one iteration early)
        live = min(dper, fper, wper)
        for name, lo, hi in PH:
            its = [i for i in range(lo, min(hi, live))]
            if not its:
                continue
            us_p = mean([x for i in its for x in di[i]])
            fb_p = 2.0 * mean([x for i in its for x in fi[i]]) * 1024.0
            wb_p = mean([x for i in its for x in wi[i]]) * 1024.0
            gbs = (fb_p + wb_p) / (us_p * 1e-6) / 1e9 if us_p > 0 else 0.0
            ph[name] = {"iterations": [its[0], its[-1]], "launches": len(its) * CALLS, "avg_us": us_p,
                        "fetch_bytes_per_launch": fb_p, "write_bytes_per_launch": wb_p, "hbm_bytes_per_launch": fb_p + wb_p,
                        "counter_GBs": gbs, "frac_of_6300_GBs_achievable": gbs / 6300.0,
                        "what": "the SAME launches in the three passes: iterations %d..%d of each of the %d calls" % (its[0], its[-1], CALLS)}
        ph["launches_per_call_trace_fetch_write_pass"] = [dper, fper, wper]
    phases[k] = ph
if phases:
    with open(os.path.join(dst, "%s_pmc_phases.json" % tag), "w") as fh:
        json.dump(phases, fh, indent=1, sort_keys=True)

# ---- VALU issue of the batched run at saturation
f = one("pmc_valu/*counter_collection.csv")
wall = None
try:
    for ln in open(os.path.join(src, "valu_wall.log")):
        if ln.startswith("B "):
            wall = float(ln.split("registrations/s")[0].split(":")[1])
except Exception:
    pass
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    regs = 2 * 256.0   # tools/gpu_batch.py 10000 1 256: one warm-up call + one timed call
    tot = collections.defaultdict(float)
    per_kernel = {}
    for k, d in agg.items():
        if d.get("SQ_ACTIVE_INST_VALU", 0.0) < 1e6:
            continue
        per_kernel[k] = {c: d[c] for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_WAVE_CYCLES") if c in d}
        for c in d:
            tot[c] += d[c]
    # SQ_ACTIVE_INST_VALU counts quad-cycles (x4 = cycles), summed over all SIMDs
    simd_us_per_reg = tot["SQ_ACTIVE_INST_VALU"] * 4.0 / SIMDS / (CLOCK_GHZ * 1e3) / regs
    out = {"command": "DISTINCT=1 CVO_HIP_GRAPH=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU ... -- python tools/gpu_batch.py 10000 1 256",
           "registrations": regs, "per_kernel_sums": per_kernel,
           "valu_active_us_per_simd_per_registration": simd_us_per_reg,
           "definition": "sum over all launches of SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz / registrations, against the wall "
                         "time per registration of the same workload without counters (tools/gpu_batch.py 10000 4 256)"}
    if wall:
        out["wall_registrations_per_s"] = wall
        out["wall_us_per_registration"] = 1e6 / wall
        out["valu_issue_frac"] = simd_us_per_reg / (1e6 / wall)
    with open(os.path.join(dst, "%s_valu.json" % tag), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)

# ---- counters of the single-stream run (head mode) and of the one-engine run, per kernel, executed launches
summary = {}
for pat, label in (("pmc_sq/*counter_collection.csv", "single_stream"), ("pmc_fetch/*counter_collection.csv", "single_stream"),
                   ("pmc_write/*counter_collection.csv", "single_stream"), ("pmc_sq_r/*counter_collection.csv", "one_engine_of_22"),
                   ("pmc_sq_acvo/*counter_collection.csv", "single_stream_acvo")):
    f = one(pat)
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if not (k.startswith("k") and "rocprim" not in k):
            continue
        for c, v in d.items():
            vv = sorted(v)
            real = [x for x in vv if x > 0.05 * vv[-1]] if vv[-1] > 0 else vv
            summary.setdefault(label, {}).setdefault(k, {})[c] = {"launches": len(v), "executed": len(real),
                                                                 "avg": sum(real) / max(len(real), 1), "max": vv[-1]}
for label in summary:
    for k, d in summary[label].items():
        if "SQ_WAIT_ANY" in d and "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"]["avg"] > 0:
            d["wait_any_over_wave_cycles"] = d["SQ_WAIT_ANY"]["avg"] / d["SQ_WAVE_CYCLES"]["avg"]
# (the resident runs: vector-issue fraction while the kernel runs -- SQ_ACTIVE_INST_VALU counts quad-cycles summed over all SIMDs,
# GRBM_GUI_ACTIVE the kernel's cycles summed over the 8 XCDs)
for label in summary:
    for k, d in summary[label].items():
        if k.startswith("kt_run") and "SQ_ACTIVE_INST_VALU" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"]["avg"] > 0:
            d["valu_active_frac"] = d["SQ_ACTIVE_INST_VALU"]["avg"] * 4.0 / (SIMDS * d["GRBM_GUI_ACTIVE"]["avg"] / 8.0)
            d["avg_us"] = d["GRBM_GUI_ACTIVE"]["avg"] / 8.0 / (CLOCK_GHZ * 1e3)
# ... and where their iterations go (tools/gpu_run_clocks.py on a -DCVO_RUN_CLOCKS build)
import re
clocks = {}
for mode in ("cvo", "acvo"):
    try:
        lines = open(os.path.join(src, "run_clocks_%s.txt" % mode)).read().splitlines()
    except Exception:
        continue
    cur = None
    for ln in lines:
        m = re.match(r"n (\d+): (\d+) iterations in ([0-9.]+) us; (\d+) runs \((\d+) declined\), (\d+) iterations inside, last record (\d+) candidates", ln)
        if m:
            cur = clocks.setdefault(mode, {}).setdefault("n_%s" % m.group(1), {"iterations": int(m.group(2)), "us": float(m.group(3)), "runs": int(m.group(4)),
                                                                               "declined": int(m.group(5)), "iterations_inside": int(m.group(6)),
                                                                               "last_record_candidates": int(m.group(7))})
            continue
        m = re.search(r"ticks of the first solver block \(2.4 per ns\): (\d+) per run-iteration", ln)
        if m and cur is not None:
            cur["ticks_per_run_iteration"] = int(m.group(1)); cur["us_per_run_iteration"] = int(m.group(1)) / 2400.0
            continue
        if cur is not None and "entry " in ln and "exch A" in ln:
            ph = {}
            for part in ln.strip().split(", "):
                mm = re.match(r"(.+?) (\d+)(?:  .*)?$", part.strip())
                if mm:
                    ph[mm.group(1).strip()] = int(mm.group(2))
            cur["ticks_by_phase"] = ph
if clocks:
    with open(os.path.join(dst, "%s_run_clocks.json" % tag), "w") as fh:
        json.dump({"what": "tools/gpu_run_clocks.py on libcvo_hip_clk.so (-DCVO_RUN_CLOCKS): ticks of the first solver block of the resident runs of one "
                           "registration (seed 20190402), 2.4 ticks per ns; entry / exit per run, the other phases per iteration inside a run", "clocks": clocks}, fh, indent=1, sort_keys=True)
if summary:
    with open(os.path.join(dst, "%s_pmc_summary.json" % tag), "w") as fh:
        json.dump(summary, fh, indent=1, sort_keys=True)

# ---- durations of the launches that did work, one registration at a time
tr = one("stats/*kernel_trace.csv")
if tr:
    d2 = collections.defaultdict(list)
    for r in csv.DictReader(open(tr)):
        d2[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    live = {}
    for k, v in d2.items():
        if "rocprim" in k or k.startswith("__amd"):
            continue
        sv = sorted(v)
        ref = sv[len(sv) // 2]
        lv = [x for x in v if x > 0.4 * ref]
        live[k] = {"launches": len(v), "avg_us": sum(v) / len(v), "live_launches": len(lv), "live_avg_us": sum(lv) / max(len(lv), 1),
                   "p50_us": ref, "total_us": sum(v)}
    with open(os.path.join(dst, "%s_kernel_live.json" % tag), "w") as fh:
        json.dump(live, fh, indent=1, sort_keys=True)
print("summaries in", dst, ":", sorted(os.listdir(dst)))
