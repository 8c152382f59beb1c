#!/usr/bin/env python3
"""What ONE large system costs per BFGS iteration, alone and next to others: the question behind the cooperative (several
workgroups per system) BFGS class.  For each atom count a synthetic system of the given kind is minimised for a fixed
number of iterations (gradient tolerance 0, so nothing converges) as 1, 8, 64, ... copies that share their term tables.
Prints one JSON line per (atoms, copies): microseconds per iteration of the slowest system (= the call), and the
inverse-Hessian bytes per second the call stands for (8 n (n + 2) bytes per iteration and system).
Usage: python tools/bench_large_systems.py [--kind dg|etk|mmff] [--atoms 300,500,1063] [--copies 1,8,64] [--iters 40]"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nvmolkit_amd import synthetic  # noqa: E402
from nvmolkit_amd.forcefield import DG, ETK, MMFF, FlatForcefieldBatch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="dg", choices=("dg", "etk", "mmff"))
ap.add_argument("--atoms", default="300,500,1063")
ap.add_argument("--copies", default="1,8,64")
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--repeat", type=int, default=2)
args = ap.parse_args()
kind = {"dg": DG, "etk": ETK, "mmff": MMFF}[args.kind]
w0, w1 = (0.7, 0.3) if kind == DG else (1.0, 1.0)
for n_atoms in (int(x) for x in args.atoms.split(",")):
    rng = np.random.default_rng(1000 + n_atoms)
    pos, groups = synthetic.random_ff_system(kind, n_atoms, rng)
    a1, flat1, g1 = synthetic.build_ff_batch_arrays(kind, [(pos, groups)])
    dim = 4 if kind == DG else 3
    n = n_atoms * dim
    for copies in (int(x) for x in args.copies.split(",")):
        a_s = np.arange(copies + 1, dtype=np.int32) * n_atoms
        sys_mol = np.zeros(copies, dtype=np.int32)
        gpu = FlatForcefieldBatch(kind, a_s, g1, system_mol=sys_mol)
        start = np.tile(flat1, copies) + np.random.default_rng(7).normal(scale=1e-3, size=copies * len(flat1))
        best = None
        for rep in range(args.repeat + 1):  # first run: warm-up (module load, scratch pools)
            p = torch.from_numpy(start).cuda()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e, st, it = gpu.minimize(p, max_iters=args.iters, grad_tol=0.0, w0=w0, w1=w1)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep > 0 and (best is None or dt < best):
                best = dt
        iters, total = int(it.max().item()), int(it.sum().item())  # (a system may meet the step-size test before the cap)
        print(json.dumps({"kind": args.kind, "atoms": n_atoms, "coordinates": n, "copies": copies, "iterations": iters,
                          "mean_iterations": round(total / copies, 1),
                          "seconds": round(best, 5), "us_per_iteration": round(best / max(iters, 1) * 1e6, 1),
                          "hessian_TB_per_s": round(8.0 * n * (n + 2) * total / best / 1e12, 4)}), flush=True)
