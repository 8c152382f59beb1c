#!/usr/bin/env python3
"""Where does an MMFF evaluation spend its time?  Minimises the same ETKDG conformers with subsets of the term groups
enabled (nvmk_ff_batch.group_mask) and prints the kernels' phase clocks (NVMK_BFGS_PROFILE=1) per subset.
Usage: python tools/probe_mmff_groups.py [--mols 300] [--iters 60]"""
import argparse
import os
import sys
from pathlib import Path

os.environ["NVMK_BFGS_PROFILE"] = "1"
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from nvmolkit_amd import synthetic  # noqa: E402
from nvmolkit_amd.embedMolecules import FlatMolecule, FlatMoleculeSet, embed_flat  # noqa: E402
from nvmolkit_amd.forcefield import MMFF, FlatForcefieldBatch, stack_molecule_tables  # noqa: E402
from nvmolkit_amd.types import CoordinateOutput  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mols", type=int, default=300)
ap.add_argument("--iters", type=int, default=60)
args = ap.parse_args()
library = synthetic.druglike_library(args.mols, seed=20260926, mean_atoms=48)
molset = FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in library])
dev = embed_flat(molset, confs_per_molecule=10, max_iterations=10, seed=1, output=CoordinateOutput.DEVICE)
torch.cuda.synchronize()
values = dev.values.torch()
batch = FlatForcefieldBatch(MMFF, dev.atom_starts.torch().cpu().numpy(), stack_molecule_tables(MMFF, [m["mmff"] for m in library]),
                            device=values.device, system_mol=dev.mol_indices.torch().to(torch.int32))
SUBSETS = [("all groups", 0x7F), ("bonded only (0-4)", 0x1F), ("pairs only (5, 6)", 0x60), ("vdW only (5)", 0x20),
           ("electrostatics only (6)", 0x40), ("bond + angle (0, 1)", 0x03), ("stretch-bend (2)", 0x04), ("oop (3)", 0x08),
           ("torsion (4)", 0x10), ("none", 0x00)]
for label, mask in SUBSETS:
    batch._c.group_mask = mask
    pos = values.reshape(-1).clone()
    sys.stderr.write(f"== {label}\n")
    sys.stderr.flush()
    batch.minimize(pos, max_iters=args.iters)
    torch.cuda.synchronize()
