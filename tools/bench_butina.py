#!/usr/bin/env python3
"""Timing of the matrix-free Butina path (BASELINE.json configs[1], second half): the N x N neighbour-count
pass and the full fused_butina call.  Usage: python tools/bench_butina.py [N ...] [--cutoff 0.3]"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from bench import SEED, synth_fingerprints  # noqa: E402
from nvmolkit_amd.clustering import fused_butina, update_neighbor_counts  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("sizes", nargs="*", type=int, default=[100_000, 1_000_000])
ap.add_argument("--cutoff", type=float, default=0.3)
ap.add_argument("--skip-butina", action="store_true")
ap.add_argument("--repeat", type=int, default=0, help="further fused_butina calls on the same set, timed one by one")
ap.add_argument("--spread", action="store_true", help="cluster centres with bit densities from 1 % to 6 % (wide popcount spread)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
for n in args.sizes:
    x = synth_fingerprints(n, 64, dev, SEED, density_range=(0.01, 0.06) if args.spread else None)
    counts = torch.zeros(n, dtype=torch.int32, device=dev)
    update_neighbor_counts(x[:4096], x[:4096], counts[:4096], 1.0 - args.cutoff)  # warm-up
    counts.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    update_neighbor_counts(x, x, counts, 1.0 - args.cutoff)
    torch.cuda.synchronize()
    t_counts = time.perf_counter() - t0
    res = {"n": n, "cutoff": args.cutoff, "neighbor_counts_s": t_counts, "pairs_per_s": n * n / t_counts,
           "max_degree": int(counts.max()), "mean_degree": float(counts.float().mean())}
    if not args.skip_butina:
        t0 = time.perf_counter()
        clusters, sizes = fused_butina(x, args.cutoff)
        t_b = time.perf_counter() - t0
        again = []
        for _ in range(args.repeat):  # (the first call grows the scratch pools: bench.py quotes the second)
            t0 = time.perf_counter()
            fused_butina(x, args.cutoff)
            again.append(time.perf_counter() - t0)
        if again:
            res["fused_butina_again_s"] = again
        res.update({"fused_butina_s": t_b, "n_clusters": len(clusters), "largest": len(clusters[0]),
                    "n_singletons": sum(1 for c in clusters if len(c) == 1)})
    print(json.dumps(res), flush=True)
