#!/usr/bin/env python3
"""Summarise a tools/profile_dense.sh output directory: per-kernel mean of every PMC counter and the
rocprofv3 --stats line of the matching kernel.  Usage: pmc_summary.py <dir> <kernel-substring> [out.json]"""
import collections
import csv
import glob
import json
import sys

d, pat = sys.argv[1], sys.argv[2]
out = {"kernel_filter": pat, "counters": {}}
for f in sorted(glob.glob(f"{d}/pmc*/**/*_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        out["counters"][c] = {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
for f in glob.glob(f"{d}/trace/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]:
            out["stats"] = {k: r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage")}
c = out["counters"]
g = lambda k: c.get(k, {}).get("mean_per_dispatch")
if g("GRBM_GUI_ACTIVE") and "stats" in out:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    out["effective_clock_GHz"] = g("GRBM_GUI_ACTIVE") / 8.0 / float(out["stats"]["AverageNs"])
if g("SQ_WAVE_CYCLES"):
    for k in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS"):
        if g(k):
            out[f"{k}/SQ_WAVE_CYCLES"] = g(k) / g("SQ_WAVE_CYCLES")
if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE"):
    out["MfmaUtil_pct"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8.0 * 1024) * 100  # 256 CU x 4 SIMD
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
