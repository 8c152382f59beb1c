#!/usr/bin/env python3
"""Occupancy over time of the fused BFGS kernels from the per-system records the library writes with NVMK_BFGS_PROFILE=1
NVMK_BFGS_TIMELINE=<file> (one line per system: kind, coordinates, first / last clock of its workgroup in 100 MHz ticks, XCC, CU,
SIMD, iterations, busy ticks).  Launches are told apart by gaps in the records; for each launch group (everything that overlaps
in time = the size classes of one minimisation stage of one batch) the script reports its span, the wave-slot occupancy
integrated over the span (waves of the systems in flight / 2048 wave slots of the chip at two waves per SIMD) and how much of the span is
"tail": time during which fewer than half of the slots are occupied.
Usage: python tools/bfgs_timeline.py FILE [--slots 2048]"""
import argparse
import gzip
import json

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("file")
ap.add_argument("--slots", type=int, default=2048)
ap.add_argument("--wave", type=int, default=176)
ap.add_argument("--wave2", type=int, default=256)
ap.add_argument("--wave8", type=int, default=656, help="smallest system of the eight-wave class (NVMK_BFGS_WAVE8); 0: the run had none")
args = ap.parse_args()
op = gzip.open if args.file.endswith(".gz") else open
rec = np.loadtxt(op(args.file, "rt"), dtype=np.int64, ndmin=2)
kind, n, t0, t1, iters = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3], rec[:, 7]
team = rec[:, 9] if rec.shape[1] > 9 else np.zeros_like(n)  # workgroups of the system's team (round 6), 0 = one workgroup
waves = np.where(team > 0, 8 * team, np.where(n <= args.wave, 1, np.where(n <= args.wave2, 2, np.where((args.wave8 > 0) & (n >= args.wave8), 8, 4))))
order = np.argsort(t0)
kind, n, t0, t1, waves, iters, team = kind[order], n[order], t0[order], t1[order], waves[order], iters[order], team[order]
# launch groups: a new group starts where a system begins after everything before it has ended (plus 20 us of slack)
groups, start, end = [], 0, t1[0]
for i in range(1, len(t0)):
    if t0[i] > end + 2000:
        groups.append((start, i))
        start = i
    end = max(end, t1[i])
groups.append((start, len(t0)))
out = {"systems": int(len(t0)), "launch_groups": len(groups), "groups": []}
tot_span = tot_busy = tot_tail = 0.0
for a, b in groups:
    s, e = t0[a:b].min(), t1[a:b].max()
    span = (e - s) * 1e-2  # us
    busy = float(((t1[a:b] - t0[a:b]) * waves[a:b]).sum()) * 1e-2  # wave-us
    # occupancy curve on a 100-us grid
    edges = np.arange(s, e + 10000, 10000)
    occ = np.zeros(len(edges) - 1)
    for w in np.unique(waves[a:b]):
        m = waves[a:b] == w
        if not m.any():
            continue
        up = np.histogram(t0[a:b][m], bins=edges)[0].cumsum()
        down = np.histogram(t1[a:b][m], bins=edges)[0].cumsum()
        occ += w * (up - down + np.histogram(t1[a:b][m], bins=edges)[0] * 0.5)
    tail = float((occ < 0.5 * args.slots).sum()) * 100.0
    kinds = {int(k): int((kind[a:b] == k).sum()) for k in np.unique(kind[a:b])}
    out["groups"].append({"kinds": kinds, "systems": int(b - a), "span_ms": span * 1e-3, "mean_occupancy": busy / (span * args.slots),
                          "tail_ms_below_half": tail * 1e-3, "mean_iterations": float(iters[a:b].mean()),
                          "longest_system_ms": float((t1[a:b] - t0[a:b]).max()) * 1e-5})
    tot_span += span
    tot_busy += busy
    tot_tail += tail
# Per size class over the whole file (round 5: what bounds the large molecules?): the classes of csrc/minimize.hip by coordinates —
# one wave (<= 176), two waves (<= 256), four waves with vectors in LDS at two workgroups per CU (A, <= 655) or one per CU (B, <= 1320),
# vectors in HBM (C).  For every class: systems, summed run time of its workgroups, the longest single system, and how much of the
# wall the class spends with fewer workgroups in flight than the chip has CUs (256) / XCDs x 8 (64): that is time in which a system's
# triangle streams at one CU's bandwidth while the rest of the chip idles — the "tail" a cooperative class would attack; a class that
# keeps >= 256 workgroups in flight for most of its wall is bound by the chip's bytes instead.
# (from round 5's eight-wave class on: 656-1067 coordinates eight waves with the vectors in LDS, beyond that in HBM; --wave8 0 gives
# the four-wave classes B (<= 1320) and C of the runs before it)
if args.wave8 > 0:
    bounds = [(args.wave, "one wave"), (args.wave2, "two waves"), (args.wave8 - 1, "four waves A"), (1067, "eight waves (LDS vectors)"),
              (1 << 30, "eight waves (HBM vectors)")]
else:
    bounds = [(args.wave, "one wave"), (args.wave2, "two waves"), (655, "four waves A"), (1320, "four waves B"), (1 << 30, "four waves C (HBM vectors)")]
# Per kind and size bin (round 6, history form): CU time (a team's workgroups all counted) and how the kernel's own phase clocks split
# it — line-search energy evaluations, gradients, the product H g (triangle pass or history product), update + direction.
if rec.shape[1] > 13:
    ph = rec[order][:, 10:14].astype(np.float64)
    busy_ticks = rec[order][:, 8].astype(np.float64)
    out["by_kind_and_size"] = []
    edges_n = [0, 176, 256, 400, 655, 799, 1100, 1500, 2047, 2900, 1 << 30]
    for k in np.unique(kind):
        for lo_n, hi_n in zip(edges_n[:-1], edges_n[1:]):
            m = (kind == k) & (n > lo_n) & (n <= hi_n)
            if not m.any():
                continue
            width = np.maximum(team[m], 1).astype(np.float64)
            cu_share = np.where(team[m] > 0, width, waves[m] / 8.0)  # fraction of a CU's wave slots x workgroups
            cu_ms = float(((t1[m] - t0[m]) * cu_share).sum()) * 1e-5
            tot = np.maximum(ph[m].sum(), 1.0)
            out["by_kind_and_size"].append({"kind": int(k), "coordinates": [int(n[m].min()), int(n[m].max())], "systems": int(m.sum()),
                                            "cu_ms": round(cu_ms, 1), "mean_iterations": round(float(iters[m].mean()), 1),
                                            "us_per_iteration": round(float(busy_ticks[m].sum()) / max(float(iters[m].sum()), 1.0) * 1e-2, 1),
                                            "phase_fractions_energy_gradient_product_update": [round(float(ph[m][:, j].sum() / tot), 3) for j in range(4)],
                                            "phase_us_per_iteration": [round(float(ph[m][:, j].sum()) / max(float(iters[m].sum()), 1.0) * 1e-2, 1) for j in range(4)]})
out["by_class"] = []
lo = 0
for hi, name in bounds:
    m = (n > lo) & (n <= hi)
    lo = hi
    if not m.any():
        continue
    a0, a1 = t0[m], t1[m]
    ev = np.concatenate([np.stack([a0, np.ones_like(a0)], 1), np.stack([a1, -np.ones_like(a1)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    level = np.cumsum(ev[:, 1])[:-1]
    dt = np.diff(ev[:, 0]).astype(np.float64)
    busy_wall = float(dt[level > 0].sum())
    out["by_class"].append({"class": name, "coordinates_min_max": [int(n[m].min()), int(n[m].max())], "systems": int(m.sum()),
                            "workgroup_ms_total": float((a1 - a0).sum()) * 1e-5, "longest_system_ms": float((a1 - a0).max()) * 1e-5,
                            "mean_system_ms": float((a1 - a0).mean()) * 1e-5, "mean_iterations": float(iters[m].mean()),
                            "wall_ms_with_any_in_flight": busy_wall * 1e-5,
                            "wall_fraction_below_256_in_flight": float(dt[(level > 0) & (level < 256)].sum()) / max(busy_wall, 1.0),
                            "wall_fraction_below_64_in_flight": float(dt[(level > 0) & (level < 64)].sum()) / max(busy_wall, 1.0),
                            "mean_in_flight": float((dt * level).sum()) / max(busy_wall, 1.0)})
# round 6: the team classes by width — systems, CU-milliseconds (run time x workgroups), iterations, inverse-Hessian bytes per second
# and CU while a system of the class runs
out["by_team_width"] = []
for w in np.unique(team):
    m = team == w
    cus = np.maximum(w, 1) if w > 0 else waves[m] / 8.0
    dur = (t1[m] - t0[m]).astype(np.float64) * 1e-5  # ms
    nbytes = 8.0 * n[m].astype(np.float64) * (n[m] + 2) * iters[m]
    out["by_team_width"].append({"workgroups_per_system": int(w), "systems": int(m.sum()), "coordinates_min_max": [int(n[m].min()), int(n[m].max())],
                                 "cu_ms_total": float((dur * cus).sum()), "longest_system_ms": float(dur.max()), "mean_iterations": float(iters[m].mean()),
                                 "inverse_hessian_TB": float(nbytes.sum()) * 1e-12,
                                 "GB_per_s_per_cu_while_running": float(nbytes.sum()) / max(float((dur * cus).sum()) * 1e-3, 1e-9) * 1e-9})
out["total_span_ms"] = tot_span * 1e-3
out["total_mean_occupancy"] = tot_busy / (tot_span * args.slots)
out["total_tail_ms_below_half"] = tot_tail * 1e-3
out["wall_ms_first_to_last"] = float(t1.max() - t0.min()) * 1e-5
print(json.dumps(out, indent=1))
