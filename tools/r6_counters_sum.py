#!/usr/bin/env python3
"""Sums of the counter passes of tools/gpu_r6_counters.sh per kernel (rocprofv3 counter_collection.csv: one row per dispatch and counter)."""
import csv, glob, json, os, sys
out = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("cvo_dev::", "")
            k = k.replace("void ", "")
            d = out.setdefault(k, {})
            c = row["Counter_Name"]
            v = d.setdefault(c, [0.0, set()])
            v[0] += float(row["Counter_Value"]); v[1].add(row["Dispatch_Id"])
res = {}
for k, d in out.items():
    r = {c: v[0] for c, v in d.items()}
    r["dispatches"] = max(len(v[1]) for v in d.values())
    if "SQ_WAVE_CYCLES" in r and r["SQ_WAVE_CYCLES"] > 0:
        wc = r["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS"):
            if c in r: r[c + "_per_wave_cycle"] = round(r[c] / wc, 4)
    if "TA_TA_BUSY" in r and r.get("GRBM_GUI_ACTIVE", 0) > 0:
        r["ta_busy_frac_of_gui_active_x_units"] = None   # (filled by the reader: TA_TA_BUSY is summed over the texture units that reported)
    res[k] = r
keep = {k: v for k, v in res.items() if k.startswith("kt_") or "k_" in k}
print(json.dumps(keep, indent=1, sort_keys=True))
