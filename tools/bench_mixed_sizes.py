#!/usr/bin/env python3
"""Conformer path on a batch that mixes drug-sized molecules with a few large ones: ETKDG + MMFF on
  (a) `--mols` molecules of the benchmark's size distribution,
  (b) the same plus `--large-frac` of them replaced by molecules of `--large-lo`..`--large-hi` atoms,
  (c) the large ones alone.
The minimiser splits every launch into size classes (nvmolkit_amd/csrc/minimize.hip): what is reported is whether the
small molecules keep their rate next to the large ones, i.e. t(b) against t(a) + t(c)."""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nvmolkit_amd import mmffOptimization, synthetic  # noqa: E402
from nvmolkit_amd.embedMolecules import FlatMolecule, FlatMoleculeSet, embed_flat  # noqa: E402
from nvmolkit_amd.types import CoordinateOutput  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mols", type=int, default=2000)
ap.add_argument("--confs", type=int, default=10)
ap.add_argument("--large-frac", type=float, default=0.01)
ap.add_argument("--large-lo", type=int, default=150)
ap.add_argument("--large-hi", type=int, default=400)
ap.add_argument("--mmff-iters", type=int, default=200)
args = ap.parse_args()

small = synthetic.druglike_library(args.mols, seed=20260926)
rng = np.random.default_rng(7)
n_large = max(1, int(round(args.mols * args.large_frac)))
large = [synthetic.druglike_molecule(rng, int(n)) for n in rng.integers(args.large_lo, args.large_hi + 1, n_large)]


def run(library, label):
    molset = FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in library])
    tables = mmffOptimization.resident_tables([m["mmff"] for m in library])
    torch.cuda.synchronize()
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        dev = embed_flat(molset, confs_per_molecule=args.confs, max_iterations=10, seed=1, output=CoordinateOutput.DEVICE)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        opt = mmffOptimization.optimize_device(tables, dev, max_iters=args.mmff_iters)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if best is None or t2 - t0 < best["seconds"]:
            best = {"set": label, "molecules": len(library), "atoms_max": max(m["embed"]["n_atoms"] for m in library),
                    "conformers": dev.num_conformers, "etkdg_s": t1 - t0, "mmff_s": t2 - t1, "seconds": t2 - t0,
                    "mols_per_s": len(library) / (t2 - t0)}
    print(json.dumps(best), flush=True)
    return best


embed_flat(FlatMoleculeSet([FlatMolecule(**small[0]["embed"])]), 1, 5)  # warm-up
a = run(small, "small only")
mixed = list(small)
for k, m in enumerate(large):
    mixed[(k * 97) % len(mixed)] = m  # scattered through the batch
b = run(mixed, f"small + {n_large} large ({args.large_lo}-{args.large_hi} atoms)")
c = run(large, "large only")
print(json.dumps({"t_mixed_over_t_small": b["seconds"] / a["seconds"],
                  "t_mixed_over_t_small_plus_t_large": b["seconds"] / (a["seconds"] + c["seconds"]),
                  "large_atoms": [m["embed"]["n_atoms"] for m in large]}))
