#!/usr/bin/env python3
"""Summarise the SQ counter passes of `tools/gpu_session.sh <name> sq_counters` (rocprofv3 --pmc, two passes per workload, counters
per kernel summed over its dispatches): ratios to SQ_WAVE_CYCLES for the BFGS kernels by force-field kind and thread count and for
the row-panel neighbour-count kernel, with the kernel-source hashes bench.py puts into its line.
Usage: python tools/sq_summary.py <session-dir>"""
import collections
import csv
import glob
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
d = sys.argv[1]


def short(name):
    m = re.search(r"(t64|t128|t256|t512)::(bfgs_kernel|bfgs_team_kernel)<(\d+)", name)
    if m:
        return f"{m.group(2)}<{('DG', 'ETK', 'MMFF', 'QUARTIC', 'UFF')[int(m.group(3))] if int(m.group(3)) < 5 else m.group(3)}> {m.group(1)}"
    if "morgan" in name and "kernel" in name:
        return "morgan_kernel"
    if "neighbor_count_panel_kernel" in name:
        return "neighbor_count_panel_kernel"
    return None


def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"{d}/{sub}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])  # (the name holds "(anonymous namespace)")
    return agg


out = {"note": "rocprofv3 --pmc (8 SQ counters per pass, two passes per workload; SQ_WAVE_CYCLES in both); counters summed over all dispatches "
               "of a kernel; SQ_* cycle counters are in quad-cycles (MI355X_MICROARCH.md), ratios are to the same pass's SQ_WAVE_CYCLES",
       "workloads": {"conformers": "tools/bench_conformers.py --mols 2000 (ETKDG x 10 + MMFF94, synthetic drug-like set)",
                     "panel": "tools/bench_butina.py 1000000 --skip-butina (1M x 1M symmetric neighbour-count pass, 2048 bit)",
                     "morgan": "tools/bench_morgan.py (the Morgan kernel per size bucket on the reference's benchmark molecules)"}}
try:
    import bench

    out["kernel_source_sha256"] = {"conformers": bench.conformer_source_digest(),
                                   "neighbour_count": bench.kernel_source_digest(("similarity_mfma.hip", "count_panel.inc", "fp4.h", "tile_maps.h"))}
except Exception as exc:  # noqa: BLE001
    out["kernel_source_sha256"] = {"error": str(exc)}
kernels = collections.defaultdict(dict)
for sub in ("sq_conf_1", "sq_conf_2", "sq_panel_1", "sq_panel_2", "sq_morgan_1", "sq_morgan_2"):
    for key, c in counters(sub).items():
        if key is None:
            continue
        w = c.get("SQ_WAVE_CYCLES", 0.0)
        row = kernels[key].setdefault(sub[-1], {})
        for k, v in c.items():
            row[k] = v
            if w and k != "SQ_WAVE_CYCLES":
                row[k + "/SQ_WAVE_CYCLES"] = v / w
        if c.get("SQ_LDS_IDX_ACTIVE"):
            row["SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        if c.get("SQ_INSTS_MFMA") and c.get("SQ_BUSY_CYCLES"):
            row["mfma_issue_share: SQ_INSTS_MFMA x 8 quad-cycles / SQ_WAVE_CYCLES"] = c["SQ_INSTS_MFMA"] * 8.0 / w if w else None
out["kernels"] = {k: {"pass_" + p: v for p, v in sorted(passes.items())} for k, passes in sorted(kernels.items())}
print(json.dumps(out, indent=1))
