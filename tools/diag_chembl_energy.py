#!/usr/bin/env python3
"""Diagnostic: which conformers of the ChEMBL block (<= 128 atoms) end an MMFF minimisation ABOVE their starting energy?
(tests/test_chembl_conformers_gpu.py::test_mmff_energies_decrease_and_equal_the_oracle_energy)"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nvmolkit_amd import mmffOptimization, synthetic  # noqa: E402
from nvmolkit_amd.embedMolecules import FlatMolecule, FlatMoleculeSet, embed_flat  # noqa: E402
from nvmolkit_amd.forcefield import MMFF, FlatForcefieldBatch, stack_molecule_tables  # noqa: E402
from nvmolkit_amd.types import CoordinateOutput  # noqa: E402
from oracle import ffc  # noqa: E402

batch_size = int(sys.argv[1]) if len(sys.argv) > 1 else -1
lib, ids = synthetic.smiles_file_library(ROOT / "tests" / "golden" / "chembl_10k.smi", max_atoms=128)
molset = FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in lib])
dev = embed_flat(molset, confs_per_molecule=10, max_iterations=10, seed=1, output=CoordinateOutput.DEVICE, batch_size=batch_size)
tables = [m["mmff"] for m in lib]
opt = mmffOptimization.optimize_device(tables, dev, max_iters=200)
a_s = dev.atom_starts.torch().cpu().numpy()
mol_of = dev.mol_indices.torch().to(torch.int32)
batch = FlatForcefieldBatch(MMFF, a_s, stack_molecule_tables(MMFF, tables), system_mol=mol_of)
e0 = batch.compute_energy(dev.values.torch().reshape(-1).contiguous()).cpu().numpy()
e1 = opt.energies.torch().cpu().numpy()
e1b = batch.compute_energy(opt.values.torch().reshape(-1).contiguous()).cpu().numpy()
conv = opt.converged.torch().cpu().numpy()
bad = np.flatnonzero(~(e1 <= e0 + 1e-9))
print("conformers", len(e0), "bad", len(bad), "nan e0", int(np.isnan(e0).sum()), "nan e1", int(np.isnan(e1).sum()))
mo = mol_of.cpu().numpy()
xyz0 = dev.values.torch().cpu().numpy()
for c in bad[:12]:
    n = a_s[c + 1] - a_s[c]
    p = xyz0[a_s[c]:a_s[c + 1]]
    d = np.linalg.norm(p[:, None] - p[None], axis=2) + np.eye(n) * 9
    cpu = ffc.Batch(MMFF, np.array([0, n]), stack_molecule_tables(MMFF, [tables[mo[c]]]))
    print(f"conf {c} mol {mo[c]} (file row {ids[mo[c]]}) atoms {n} e0 {e0[c]:.6g} e1 {e1[c]:.6g} recomputed {e1b[c]:.6g} converged {conv[c]} "
          f"min dist start {d.min():.4f} oracle e0 {cpu.energy(p.reshape(-1))[0]:.6g}")
    # the same start through the oracle's BFGS, and alone through the GPU minimiser with 0 / 1 / 2 / 5 / 200 iterations
    xo, eo, so, io = cpu.minimize(p.reshape(-1), max_iters=200)
    print(f"   oracle BFGS: energy {eo[0]:.8g} status {so[0]} iterations {io[0]}")
    one = FlatForcefieldBatch(MMFF, np.array([0, n], dtype=np.int32), stack_molecule_tables(MMFF, [tables[mo[c]]]))
    for it in (0, 1, 2, 5, 20, 200):
        q = torch.from_numpy(p.reshape(-1).copy()).cuda()
        e, st, k = one.minimize(q, max_iters=it)
        print(f"   gpu alone, max_iters {it}: energy {float(e[0]):.8g} status {int(st[0])} iterations {int(k[0])} energy kernel at the result {float(one.compute_energy(q)[0]):.8g}")
    per = np.array([float(FlatForcefieldBatch(MMFF, np.array([0, n], dtype=np.int32), [(np.array([0, len(i)], dtype=np.int32), i, pp) if g == gg else (np.array([0, 0], dtype=np.int32), i[:0], pp[:0])
                                                                                       for gg, (i, pp) in enumerate(tables[mo[c]])]).compute_energy(torch.from_numpy(p.reshape(-1).copy()).cuda())[0]) for g in range(7)])
    print("   per-group energies at the start:", np.round(per, 5).tolist())
    q = torch.from_numpy(p.reshape(-1).copy()).cuda()
    one.minimize(q, max_iters=200)
    per1 = np.array([float(FlatForcefieldBatch(MMFF, np.array([0, n], dtype=np.int32), [(np.array([0, len(i)], dtype=np.int32), i, pp) if g == gg else (np.array([0, 0], dtype=np.int32), i[:0], pp[:0])
                                                                                        for gg, (i, pp) in enumerate(tables[mo[c]])]).compute_energy(q)[0]) for g in range(7)])
    print("   per-group energies at the result:", np.round(per1, 5).tolist(), "largest move", float(np.abs(q.cpu().numpy() - p.reshape(-1)).max()))
