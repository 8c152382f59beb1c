#!/usr/bin/env python3
"""Throughput of the conformer path on synthetic flattened molecules (BASELINE.json configs[2] / configs[3] shape):
ETKDG with numConfs conformers per molecule, then MMFF94 optimisation of every conformer.
Usage: python tools/bench_conformers.py [--mols 1000] [--confs 10] [--mean-atoms 48]
Multi-GPU (configs[3]): python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
  tools/bench_conformers.py --mols M   -- `--mols` is the WHOLE batch; molecules are dealt to ranks by cost
  (nvmolkit_amd.distributed.shard_molecules_by_cost), no data-path collective; times are the max over ranks."""
import os
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nvmolkit_amd.embedMolecules import STAGE_NAMES, FlatMolecule, FlatMoleculeSet, embed_flat  # noqa: E402
from nvmolkit_amd.forcefield import MMFF, FlatForcefieldBatch  # noqa: E402
from nvmolkit_amd import synthetic as util  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mols", type=int, default=1000)
ap.add_argument("--confs", type=int, default=10)
ap.add_argument("--mean-atoms", type=int, default=48)
ap.add_argument("--batch-size", type=int, default=4096)
ap.add_argument("--mmff-iters", type=int, default=200)
ap.add_argument("--batches-per-gpu", type=int, default=1)
args = ap.parse_args()
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
if world > 1:
    import torch.distributed as dist

    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl")
rng = np.random.default_rng(20260926)
all_sizes = np.clip(rng.normal(args.mean_atoms, 12, size=args.mols).round().astype(int), 12, 96)
if world > 1:
    from nvmolkit_amd.distributed import shard_molecules_by_cost

    sizes = all_sizes[shard_molecules_by_cost(all_sizes, world, rank)]
    rng = np.random.default_rng(20260926 + 1 + rank)
else:
    sizes = all_sizes
args.mols = len(sizes)
t0 = time.perf_counter()
mols = [FlatMolecule(**util.synthetic_embed_molecule(rng, int(n), with_etk=True)[0]) for n in sizes]
molset = FlatMoleculeSet(mols)
t_prep = time.perf_counter() - t0
torch.cuda.synchronize()
t0 = time.perf_counter()
res = embed_flat(molset, confs_per_molecule=args.confs, max_iterations=10, batch_size=args.batch_size, enforce_chirality=False, seed=1,
                 batches_per_gpu=args.batches_per_gpu)
torch.cuda.synchronize()
t_embed = time.perf_counter() - t0
n_conf = int(res.conf_counts.sum())

# MMFF: one synthetic term table per molecule, shared by its conformers (system_mol)
t0 = time.perf_counter()
mm = [util.random_ff_system(MMFF, int(n), rng) for n in sizes]
a_s, flat, groups = util.build_ff_batch_arrays(MMFF, mm)
t_prep2 = time.perf_counter() - t0
total_iters = 0
t_mmff = 0.0
done = 0
for lo in range(0, args.mols, max(1, args.batch_size // args.confs)):
    hi = min(args.mols, lo + max(1, args.batch_size // args.confs))
    sub = [mm[i] for i in range(lo, hi) for _ in range(args.confs)]
    sa, sf, sg = util.build_ff_batch_arrays(MMFF, sub)
    noise = rng.normal(scale=0.05, size=sf.shape)
    pos = torch.from_numpy(sf + noise).cuda()
    batch = FlatForcefieldBatch(MMFF, sa, sg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e, st, it = batch.minimize(pos, max_iters=args.mmff_iters, grad_tol=1e-4)
    torch.cuda.synchronize()
    t_mmff += time.perf_counter() - t0
    total_iters += int(it.sum())
    done += len(sub)
if world > 1:  # whole-job numbers: sums of work, max of time
    t = torch.tensor([t_embed, t_mmff], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([n_conf, done, total_iters, args.mols], dtype=torch.float64, device="cuda")
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    t_embed, t_mmff = float(t[0]), float(t[1])
    n_conf, done, total_iters, args.mols = (int(x) for x in c.tolist())
    dist.barrier()
    dist.destroy_process_group()
    if rank != 0:
        sys.exit(0)
print(json.dumps({
    "n_gpus": world, "mols": args.mols, "confs_per_mol": args.confs, "mean_atoms": float(all_sizes.mean()),
    "etkdg_s": t_embed, "etkdg_conformers": n_conf, "etkdg_confs_per_s": n_conf / t_embed,
    "etkdg_stage_failures": dict(zip(STAGE_NAMES, res.stage_failures.tolist())),
    "mmff_s": t_mmff, "mmff_conformers": done, "mmff_confs_per_s": done / t_mmff, "mmff_bfgs_iterations": total_iters,
    "mols_per_s_etkdg_plus_mmff": args.mols / (t_embed + t_mmff), "host_prep_s": t_prep + t_prep2}))
