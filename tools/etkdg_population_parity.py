#!/usr/bin/env python3
"""ETKDG on the GPU against the C oracle's restatement of the whole stage pipeline, per POPULATION, at a size where the
sampling noise is small: same molecules, same seed, same scheduler and start coordinates on both sides.  Individual
200-400 iteration minimisations amplify last-digit differences into different (equally valid) embeddings, so what can agree
is the number of conformers, the failures per stage and the distribution of the conformers' bound violations.
(tests/test_config_size_gpu.py::test_etkdg_pipeline_matches_oracle_pipeline_statistically is the 96-molecule version with
test tolerances; this prints the numbers.)  Usage: python tools/etkdg_population_parity.py [--mols 1500] [--confs 4]"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

from nvmolkit_amd import synthetic  # noqa: E402
from nvmolkit_amd.embedMolecules import FlatMolecule, FlatMoleculeSet, embed_flat  # noqa: E402
from oracle import ffc  # noqa: E402
import oracle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mols", type=int, default=1500)
ap.add_argument("--confs", type=int, default=4)
ap.add_argument("--seed", type=int, default=3)
args = ap.parse_args()

lib = synthetic.druglike_library(args.mols, seed=9, processes=8)
mols = [FlatMolecule(**m["embed"]) for m in lib]
t = time.perf_counter()
gpu = embed_flat(FlatMoleculeSet(mols), confs_per_molecule=args.confs, max_iterations=10, seed=args.seed)
t_gpu = time.perf_counter() - t
t = time.perf_counter()
coords, counts, slots, fails, _ = ffc.etkdg_embed(mols, confs_per_molecule=args.confs, max_iterations=10, seed=args.seed, batch_size=16384)
t_cpu = time.perf_counter() - t


def violations(get, n_confs):
    worst = []
    for m, mol in enumerate(lib):
        pairs, lb, ub = mol["bounds"]
        for k in range(int(n_confs[m])):
            p = get(m, k)
            d = np.linalg.norm(p[pairs[:, 0]] - p[pairs[:, 1]], axis=1)
            worst.append(float(np.max(np.maximum(np.maximum(lb - d, d - ub), 0.0) / ub)))
    return np.array(worst)


def cpu_conf(m, k):
    n = lib[m]["embed"]["n_atoms"]
    return coords[slots[m] + 3 * n * k: slots[m] + 3 * n * (k + 1)].reshape(n, 3)


vg = violations(lambda m, k: gpu.conformers(m)[k].cpu().numpy(), gpu.conf_counts)
vc = violations(cpu_conf, counts)


def ks(a, b):  # two-sample Kolmogorov-Smirnov statistic
    grid = np.sort(np.concatenate([a, b]))
    return float(np.max(np.abs(np.searchsorted(np.sort(a), grid, side="right") / len(a) - np.searchsorted(np.sort(b), grid, side="right") / len(b))))


q = (10, 50, 90, 99)
same_counts = float(np.mean(np.asarray(gpu.conf_counts) == np.asarray(counts)))
print(json.dumps({
    "molecules": args.mols, "confs_per_molecule": args.confs, "seed": args.seed,
    "gpu_seconds": t_gpu, "oracle_seconds": t_cpu, "oracle_threads": oracle.num_threads(),
    "conformers": {"gpu": int(np.sum(gpu.conf_counts)), "oracle": int(np.sum(counts)),
                   "relative_difference": float(abs(int(np.sum(gpu.conf_counts)) - int(np.sum(counts))) / max(int(np.sum(counts)), 1)),
                   "molecules_with_the_same_count": same_counts},
    "stage_failures": {"gpu": [int(x) for x in np.asarray(gpu.stage_failures).reshape(-1)],
                       "oracle": [int(x) for x in np.asarray(fails).reshape(-1)]},
    "worst_relative_bound_violation_per_conformer": {
        "percentiles": list(q), "gpu": [float(np.percentile(vg, x)) for x in q], "oracle": [float(np.percentile(vc, x)) for x in q],
        "max": {"gpu": float(vg.max()), "oracle": float(vc.max())}, "kolmogorov_smirnov": ks(vg, vc),
        "ks_5_percent_critical": float(1.36 * np.sqrt((len(vg) + len(vc)) / (len(vg) * len(vc))))},
}))
