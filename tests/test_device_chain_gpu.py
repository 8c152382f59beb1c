"""DEVICE-output chaining: ETKDG -> Device3DResult -> MMFF / UFF minimisation without a host round trip
(reference nvmolkit/tests/test_embed_molecules.py DEVICE cases,
test_mmff_optimization.py / test_uff_optimization.py DEVICE cases).  Chemistry-free: molecules are the synthetic
chains of tests/util.py and the MMFF / UFF tables are synthetic tables for the same atom counts."""

import numpy as np
import pytest
import torch

from oracle import ff as off
from nvmolkit_amd import mmffOptimization, uffOptimization
from nvmolkit_amd.embedMolecules import FlatMolecule, FlatMoleculeSet, embed_flat
from nvmolkit_amd.forcefield import MMFF, UFF, FlatForcefieldBatch
from nvmolkit_amd.types import CoordinateOutput, Device3DResult, HardwareOptions
from tests import util

pytestmark = pytest.mark.gpu

SIZES = [6, 9, 12, 7]


def embed(confs=3, seed=5):
    rng = np.random.default_rng(seed)
    mols = [FlatMolecule(**util.synthetic_embed_molecule(rng, n, True)[0]) for n in SIZES]
    molset = FlatMoleculeSet(mols)
    flat = embed_flat(molset, confs_per_molecule=confs, max_iterations=20, enforce_chirality=False, seed=3)
    # (two embedding runs are not bit-identical: gradients are accumulated with floating-point atomics, so the
    # DEVICE form is derived from the SAME run here; the output=DEVICE switch is covered in the first test)
    return flat, flat.to_device_result(), molset


def test_embed_device_output_equals_flat_output():
    flat, dev, molset = embed()
    direct = embed_flat(molset, confs_per_molecule=3, max_iterations=20, enforce_chirality=False, seed=3,
                        output=CoordinateOutput.DEVICE)
    assert isinstance(direct, Device3DResult) and direct.num_conformers == dev.num_conformers
    assert torch.equal(direct.atom_starts.torch(), dev.atom_starts.torch())
    assert torch.equal(direct.mol_indices.torch(), dev.mol_indices.torch())
    assert isinstance(dev, Device3DResult) and dev.n_mols == len(SIZES) and dev.energies is None
    assert dev.num_conformers == int(flat.conf_counts.sum())
    per = dev.per_molecule()
    for m, n in enumerate(SIZES):
        assert len(per[m]) == flat.conf_counts[m]
        for k, view in enumerate(per[m]):
            assert view.shape == (n, 3)
            assert torch.equal(view, flat.conformers(m)[k])       # same seed, same coordinates, compacted on the GPU
    dense = dev.dense()
    assert dense.values.shape == (len(SIZES), int(flat.conf_counts.max()), max(SIZES), 3)
    assert int(dense.conf_mask.sum()) == dev.num_conformers
    assert dev.mol_indices.torch().tolist() == sorted(dev.mol_indices.torch().tolist())


@pytest.mark.parametrize("kind", [MMFF, UFF])
def test_device_chain_matches_flat_minimisation(kind):
    _, dev, _ = embed()
    rng = np.random.default_rng(17)
    tables = [util.random_ff_system(kind, n, rng)[1] for n in SIZES]
    mod = mmffOptimization if kind == MMFF else uffOptimization
    before = dev.values.torch().clone()
    out = mod.optimize_device(tables, dev, max_iters=200)
    assert isinstance(out, Device3DResult) and out.num_conformers == dev.num_conformers and out.n_mols == dev.n_mols
    assert torch.equal(dev.values.torch(), before)                       # the input result is left untouched
    assert out.energies.torch().shape == (dev.num_conformers,) and out.converged.torch().dtype == torch.int8
    assert torch.equal(out.atom_starts.torch(), dev.atom_starts.torch())
    # the same conformers minimised as an ordinary flat batch with per-system tables: identical arithmetic
    mols = dev.mol_indices.torch().tolist()
    systems = [(None, tables[m]) for m in mols]
    starts = dev.atom_starts.torch().cpu().numpy()
    layout = off.LAYOUT[kind]
    groups = []
    for g, (n_idx, n_par) in enumerate(layout):
        st = np.zeros(len(systems) + 1, dtype=np.int32)
        for s, (_, t) in enumerate(systems):
            st[s + 1] = st[s] + len(t[g][0])
        groups.append((st, np.concatenate([t[g][0].reshape(-1, n_idx) for _, t in systems]),
                       np.concatenate([t[g][1].reshape(-1, n_par) for _, t in systems])))
    pos = before.reshape(-1).clone()
    energies, converged = mod.optimize_flat(starts, groups, pos, max_iters=200)
    assert torch.equal(out.values.torch().reshape(-1), pos)
    assert torch.equal(out.energies.torch(), energies)
    assert torch.equal(out.converged.torch().bool(), converged)
    # energies are those of the returned coordinates, and minimisation did not raise them
    batch = FlatForcefieldBatch(kind, starts, groups)
    np.testing.assert_allclose(batch.compute_energy(pos).cpu().numpy(), energies.cpu().numpy(), rtol=1e-9, atol=1e-9)
    assert (energies.cpu().numpy() <= batch.compute_energy(before.reshape(-1).contiguous()).cpu().numpy() + 1e-9).all()


def test_device_chain_rejects_mismatched_tables():
    _, dev, _ = embed(confs=1)
    with pytest.raises(ValueError):
        mmffOptimization.optimize_device([], dev)


class FakeConformer:
    def __init__(self, cid, xyz):
        self.cid, self.xyz = cid, np.array(xyz, dtype=np.float64)

    def GetId(self):
        return self.cid

    def GetPositions(self):
        return self.xyz

    def SetPositions(self, xyz):
        self.xyz = np.array(xyz, dtype=np.float64)


class FakeMol:
    """Duck-typed stand-in for the four RDKit calls the conformer driver makes."""

    def __init__(self, confs):
        self.confs = {c.GetId(): c for c in confs}

    def GetNumAtoms(self):
        return len(next(iter(self.confs.values())).xyz)

    def GetConformers(self):
        return list(self.confs.values())

    def GetConformer(self, cid):
        return self.confs[cid]


@pytest.mark.parametrize("kind", [MMFF, UFF])
def test_rdkit_conformer_driver_on_duck_typed_molecules(kind):
    """optimize_rdkit_conformers: flatten once per molecule, batches of `batchSize`, write-back and DEVICE output
    (src/minimizer/bfgs_common.cpp:42-105, bfgs_mmff.cpp:139-328) — exercised without RDKit."""
    from nvmolkit_amd._rdkit_confs import optimize_rdkit_conformers

    rng = np.random.default_rng(23)
    systems = [util.random_ff_system(kind, n, rng) for n in (5, 8, 6)]
    mols = [FakeMol([FakeConformer(10 + c, p[:, :3] + 0.05 * rng.normal(size=p[:, :3].shape)) for c in range(k)])
            for (p, _), k in zip(systems, (2, 3, 1))]
    start = [[c.xyz.copy() for c in m.GetConformers()] for m in mols]
    calls = []

    def flatten(mi, cid):
        calls.append((mi, cid))
        return systems[mi][1]

    opts = HardwareOptions(batchSize=4)                     # 6 conformers -> two launches, molecule 1 straddles them
    dev = optimize_rdkit_conformers(kind, mols, flatten, 200, 1e-4, opts, CoordinateOutput.DEVICE, -1)
    assert calls == [(0, 10), (1, 10), (2, 10)]             # once per molecule, on its first conformer
    assert isinstance(dev, Device3DResult) and dev.num_conformers == 6 and dev.n_mols == 3
    assert dev.mol_indices.torch().tolist() == [0, 0, 1, 1, 1, 2] and dev.conf_indices.torch().tolist() == [0, 1, 0, 1, 2, 0]
    for m in mols:                                          # DEVICE mode does not touch the conformers
        for c, s in zip(m.GetConformers(), start[mols.index(m)]):
            assert np.array_equal(c.xyz, s)
    energies = optimize_rdkit_conformers(kind, mols, flatten, 200, 1e-4, opts, CoordinateOutput.RDKIT_CONFORMERS, -1)
    assert [len(e) for e in energies] == [2, 3, 1]
    flat_e = [e for per in energies for e in per]
    np.testing.assert_allclose(flat_e, dev.energies.torch().cpu().numpy(), rtol=1e-12)
    per = dev.per_molecule()
    for mi, m in enumerate(mols):
        for k, c in enumerate(m.GetConformers()):
            assert np.array_equal(c.xyz, per[mi][k].cpu().numpy())           # written back == DEVICE values
            e_oracle = off.system_energy(kind, c.xyz, systems[mi][1])
            assert e_oracle == pytest.approx(energies[mi][k], rel=1e-9, abs=1e-9)
    with pytest.raises(ValueError, match="targetGpu"):
        optimize_rdkit_conformers(kind, mols, flatten, 10, 1e-4, HardwareOptions(gpuIds=[0]), CoordinateOutput.DEVICE, 3)


def test_resident_term_tables_give_the_same_minimisation():
    """MoleculeTermTables (tables uploaded and pair-ordered once) against per-call upload: same bits."""
    from nvmolkit_amd import mmffOptimization, synthetic
    from nvmolkit_amd.embedMolecules import FlatMolecule, FlatMoleculeSet, embed_flat
    from nvmolkit_amd.types import CoordinateOutput

    lib = synthetic.druglike_library(12, seed=21, mean_atoms=30, processes=1)
    tables = [m["mmff"] for m in lib]
    dev = embed_flat(FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in lib]), confs_per_molecule=3, max_iterations=10, seed=2,
                     output=CoordinateOutput.DEVICE)
    a = mmffOptimization.optimize_device(tables, dev, max_iters=60)
    resident = mmffOptimization.resident_tables(tables)
    b = mmffOptimization.optimize_device(resident, dev, max_iters=60)
    c = mmffOptimization.optimize_device(resident, dev, max_iters=60)      # and again from the same resident tables
    assert torch.equal(a.values.torch(), b.values.torch()) and torch.equal(a.energies.torch(), b.energies.torch())
    assert torch.equal(b.values.torch(), c.values.torch())
    with pytest.raises(ValueError, match="expected term tables"):
        mmffOptimization.optimize_device(mmffOptimization.resident_tables(tables[:5]), dev)
