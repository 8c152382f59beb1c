#!/usr/bin/env python3
"""ETKDG throughput against attempts per launch and concurrent batches (one library, several settings).
Usage: python tools/sweep_embed_batch.py [--mols 10000]"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from nvmolkit_amd import mmffOptimization, synthetic  # noqa: E402
from nvmolkit_amd.embedMolecules import FlatMolecule, FlatMoleculeSet, embed_flat  # noqa: E402
from nvmolkit_amd.types import CoordinateOutput  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mols", type=int, default=10000)
args = ap.parse_args()
library = synthetic.druglike_library(args.mols, seed=20260926, mean_atoms=48)
molset = FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in library])
tables = mmffOptimization.resident_tables([m["mmff"] for m in library])
embed_flat(FlatMoleculeSet([FlatMolecule(**library[0]["embed"])]), 1, 5)
torch.cuda.synchronize()
for batch, bpg in ((16384, 1), (8192, 2), (16384, 2), (12288, 2), (24576, 1), (8192, 3), (32768, 2), (16384, 1)):
    t0 = time.perf_counter()
    dev = embed_flat(molset, confs_per_molecule=10, max_iterations=10, batch_size=batch, seed=1, output=CoordinateOutput.DEVICE,
                     batches_per_gpu=bpg)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(json.dumps({"batch_size": batch, "batches_per_gpu": bpg, "etkdg_s": t, "conformers": dev.num_conformers,
                      "conformers_per_s": dev.num_conformers / t}), flush=True)
t0 = time.perf_counter()
opt = mmffOptimization.optimize_device(tables, dev, max_iters=200)
torch.cuda.synchronize()
print(json.dumps({"mmff_s": time.perf_counter() - t0}))
