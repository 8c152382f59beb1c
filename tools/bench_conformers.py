#!/usr/bin/env python3
"""Throughput of the conformer path on the synthetic drug-like set (BASELINE.json configs[2] / configs[3] shape):
ETKDG with numConfs conformers per molecule, DEVICE-chained into an MMFF94 optimisation of every conformer.
Usage: python tools/bench_conformers.py [--mols 1000] [--confs 10] [--mean-atoms 48] [--mmff-iters 200]
Multi-GPU (configs[3]): python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
  tools/bench_conformers.py --mols M   -- `--mols` is the WHOLE batch; molecules are dealt to ranks by cost
  (nvmolkit_amd.distributed.shard_molecules_by_cost), no data-path collective; times are the max over ranks."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nvmolkit_amd import mmffOptimization  # noqa: E402
from nvmolkit_amd import synthetic  # noqa: E402
from nvmolkit_amd.embedMolecules import STAGE_NAMES, FlatMolecule, FlatMoleculeSet, embed_flat  # noqa: E402
from nvmolkit_amd.types import CoordinateOutput  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mols", type=int, default=1000)
ap.add_argument("--confs", type=int, default=10)
ap.add_argument("--mean-atoms", type=int, default=48)
ap.add_argument("--batch-size", type=int, default=-1)
ap.add_argument("--mmff-iters", type=int, default=200)
ap.add_argument("--batches-per-gpu", type=int, default=-1)
ap.add_argument("--repeat", type=int, default=1, help="timed repetitions (the best is reported)")
ap.add_argument("--set", default="synthetic", choices=("synthetic", "chembl"),
                help="synthetic: generated drug-like graphs (clipped N(48, 12) atoms); chembl: the topologies of tests/golden/chembl_10k.smi "
                "(the reference's benchmarks/data/chembl_10k.smi) with explicit hydrogens and generic parameters (synthetic.graph_molecule)")
ap.add_argument("--max-atoms", type=int, default=128, help="chembl: molecules with more atoms (hydrogens included) are left out")
ap.add_argument("--end-to-end", action="store_true", help="after the timed repetitions on resident tables, run the job once more from the "
                "per-molecule host arrays with the table assembly inside the clock (what bench.py times)")
ap.add_argument("--cache", default="", help="directory for the generated molecule library (pickle): A/B runs of several builds in one "
                "session then generate it once")
args = ap.parse_args()
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
if world > 1:
    import torch.distributed as dist

    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl")
t0 = time.perf_counter()
def load_library():
    import pickle

    tag = f"druglike_{args.mols}_{args.mean_atoms}" if args.set == "synthetic" else f"chembl_{args.mols}_{args.max_atoms}"
    path = Path(args.cache) / f"{tag}.pkl" if args.cache else None
    if path is not None and path.exists():
        with open(path, "rb") as f:
            return pickle.load(f)
    if args.set == "chembl":
        lib, _ = synthetic.smiles_file_library(ROOT / "tests" / "golden" / "chembl_10k.smi", n_mols=args.mols, max_atoms=args.max_atoms)
    else:
        lib = synthetic.druglike_library(args.mols, seed=20260926, mean_atoms=args.mean_atoms)
    if path is not None and rank == 0:
        path.parent.mkdir(parents=True, exist_ok=True)
        with open(str(path) + ".tmp", "wb") as f:
            pickle.dump(lib, f, protocol=pickle.HIGHEST_PROTOCOL)
        os.replace(str(path) + ".tmp", path)
    return lib


library = load_library()
if world > 1:
    from nvmolkit_amd.distributed import shard_molecules_by_cost

    mine = shard_molecules_by_cost(np.array([m["embed"]["n_atoms"] for m in library]), world, rank)
    library = [library[i] for i in mine]
n_mols = len(library)
molset = FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in library])
tables = mmffOptimization.resident_tables([m["mmff"] for m in library])  # term tables resident before the timed region
torch.cuda.synchronize()
t_prep = time.perf_counter() - t0
embed_flat(FlatMoleculeSet([FlatMolecule(**library[0]["embed"])]), 1, 5)  # warm-up: module load, allocator pools
torch.cuda.synchronize()
from nvmolkit_amd import _native  # noqa: E402

stats = torch.zeros(64, dtype=torch.int64, device="cuda")  # the kernels' own counters: systems, iterations, inverse-Hessian bytes
_native.check(_native.lib().nvmk_bfgs_set_stats(stats.data_ptr()))
best = None
for _ in range(args.repeat):
    t0 = time.perf_counter()
    fails_before = None
    dev = embed_flat(molset, confs_per_molecule=args.confs, max_iterations=10, batch_size=args.batch_size, seed=1,
                     output=CoordinateOutput.DEVICE, batches_per_gpu=args.batches_per_gpu)
    torch.cuda.synchronize()
    t_embed = time.perf_counter() - t0
    t0 = time.perf_counter()
    opt = mmffOptimization.optimize_device(tables, dev, max_iters=args.mmff_iters)
    torch.cuda.synchronize()
    t_mmff = time.perf_counter() - t0
    if best is None or t_embed + t_mmff < best[0] + best[1]:
        best = (t_embed, t_mmff, dev.num_conformers, int(opt.converged.torch().sum().item()))
t_embed, t_mmff, n_conf, n_converged = best
_native.check(_native.lib().nvmk_bfgs_set_stats(None))
t_end_to_end = e_molset = e_wait = float("nan")
molset_timings = {}
if args.end_to_end:
    # the same job from the per-molecule host arrays (what bench.py times): table assembly inside the clock, MMFF tables under ETKDG
    torch.cuda.synchronize()
    e0 = time.perf_counter()
    molset2 = FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in library])
    e_molset = time.perf_counter() - e0
    molset_timings = dict(molset2.timings)
    pending = mmffOptimization.resident_tables([m["mmff"] for m in library], wait=False, after=molset2)
    dev2 = embed_flat(molset2, confs_per_molecule=args.confs, max_iterations=10, batch_size=args.batch_size, seed=1,
                      output=CoordinateOutput.DEVICE, batches_per_gpu=args.batches_per_gpu)
    torch.cuda.synchronize()
    e1 = time.perf_counter()
    tables2 = pending.result()
    e_wait = time.perf_counter() - e1
    mmffOptimization.optimize_device(tables2, dev2, max_iters=args.mmff_iters)
    torch.cuda.synchronize()
    t_end_to_end = time.perf_counter() - e0
    del molset2, tables2, dev2
st = stats.cpu().numpy().reshape(8, 8) // max(args.repeat, 1)
bfgs = {name: {"systems": int(st[k, 0]), "iterations": int(st[k, 1]), "algorithmic_bytes": int(st[k, 2]), "energy_evaluations": int(st[k, 3]),
               "hbm_requested_bytes": int(st[k, 4]), "packed_triangle_bytes_of_the_same_iterations": int(st[k, 5]),
               "minimisations_in_history_form": int(st[k, 6])} for k, name in ((0, "dg"), (1, "etk"), (2, "mmff"))}
if world > 1:  # whole-job numbers: sums of work, max of time
    t = torch.tensor([t_embed, t_mmff], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([n_conf, n_converged, n_mols], dtype=torch.float64, device="cuda")
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    t_embed, t_mmff = float(t[0]), float(t[1])
    n_conf, n_converged, n_mols = (int(x) for x in c.tolist())
    dist.barrier()
    dist.destroy_process_group()
    if rank != 0:
        sys.exit(0)
sizes = np.array([m["embed"]["n_atoms"] for m in library])
print(json.dumps({
    "set": args.set, "atoms_percentiles_5_25_50_75_95_max": [int(x) for x in np.percentile(sizes, [5, 25, 50, 75, 95, 100])],
    "molecules_with_10_conformers": None if world > 1 else int((np.bincount(dev.mol_indices.torch().cpu().numpy(), minlength=n_mols) >= args.confs).sum()),
    "n_gpus": world, "mols": n_mols, "confs_per_mol": args.confs,
    "mean_atoms": float(np.mean([m["embed"]["n_atoms"] for m in library])),
    "etkdg_s": t_embed, "etkdg_conformers": n_conf, "etkdg_confs_per_s": n_conf / t_embed,
    "mmff_s": t_mmff, "mmff_confs_per_s": n_conf / t_mmff, "mmff_max_iters": args.mmff_iters,
    "mmff_converged_frac": n_converged / max(n_conf, 1),
    "mols_per_s_etkdg_plus_mmff": n_mols / (t_embed + t_mmff), "host_prep_s": t_prep,
    "end_to_end_s": t_end_to_end, "mols_per_s_end_to_end": (n_mols if world == 1 else float("nan")) / t_end_to_end,
    "table_assembly_host_s": e_molset, "table_assembly_steps": molset_timings, "mmff_tables_wait_s": e_wait, "bfgs": bfgs}))
