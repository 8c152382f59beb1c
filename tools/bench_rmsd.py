#!/usr/bin/env python3
"""Timing of the conformer RMSD matrix + greedy RMS pruning on synthetic conformer sets.
Usage: python tools/bench_rmsd.py [--mols 1000] [--confs 50] [--atoms 48]"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from nvmolkit_amd.conformerRmsd import conformer_rms_matrix_flat, prune_conformers  # noqa: E402
from nvmolkit_amd.types import Device3DResult  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mols", type=int, default=1000)
ap.add_argument("--confs", type=int, default=50)
ap.add_argument("--atoms", type=int, default=48)
args = ap.parse_args()
g = torch.Generator(device="cuda").manual_seed(7)
base = torch.randn((args.mols, 1, args.atoms, 3), dtype=torch.float64, device="cuda", generator=g) * 3.0
coords = base + 0.4 * torch.randn((args.mols, args.confs, args.atoms, 3), dtype=torch.float64, device="cuda", generator=g)
mols = [coords[m].contiguous() for m in range(args.mols)]
conformer_rms_matrix_flat(mols[:8])  # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
mats = conformer_rms_matrix_flat(mols)
torch.cuda.synchronize()
t_rms = time.perf_counter() - t0
pairs = sum(int(m.numel()) for m in mats)
n_conf = args.mols * args.confs
res = Device3DResult(coords.reshape(-1, 3).contiguous(),
                     (torch.arange(n_conf + 1, device="cuda", dtype=torch.int32) * args.atoms),
                     torch.arange(args.mols, device="cuda", dtype=torch.int32).repeat_interleave(args.confs),
                     torch.arange(args.confs, device="cuda", dtype=torch.int32).repeat(args.mols), 0, args.mols)
t0 = time.perf_counter()
pruned = prune_conformers(res, 0.5)
torch.cuda.synchronize()
t_prune = time.perf_counter() - t0
bytes_read = pairs * 2 * args.atoms * 3 * 8
print(json.dumps({"mols": args.mols, "confs": args.confs, "atoms": args.atoms, "pairs": pairs, "rmsd_s": t_rms,
                  "pairs_per_s": pairs / t_rms, "coordinate_GBps_nominal": bytes_read / t_rms / 1e9,
                  "rmsd_plus_prune_s": t_prune, "kept": int(pruned.mol_indices.torch().numel())}))
