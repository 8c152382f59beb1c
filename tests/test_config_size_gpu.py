"""The HIP path at BASELINE.json's configuration sizes (configs[0] and configs[2] had no -m gpu test).

configs[0]: 10 000 molecules -> Morgan r = 2 / 2048 bit -> 10k x 10k Tanimoto, bit-exact against oracle-Morgan ->
oracle-similarity (the shape of the reference's tests/integration/test_fp_sim_workflow.cpp, RDKit replaced by seeded
graphs).  configs[2]: 10 000 drug-like molecules x 10 conformers, ETKDG chained on the device into MMFF94: size-
independent properties on all of it, the C oracle's pipeline on a slice of it."""

import numpy as np
import pytest
import torch

import oracle
from nvmolkit_amd import mmffOptimization, synthetic
from nvmolkit_amd.embedMolecules import FlatMolecule, FlatMoleculeSet, embed_flat
from nvmolkit_amd.fingerprints import MorganFingerprintGenerator
from nvmolkit_amd.forcefield import MMFF, FlatForcefieldBatch, stack_molecule_tables
from nvmolkit_amd.similarity import crossTanimotoSimilarity
from nvmolkit_amd.types import CoordinateOutput
from oracle import ffc
from tests import util

pytestmark = pytest.mark.gpu


def test_cfg1_morgan_to_similarity_chain_10k():
    n = 10_000
    mols = util.random_molecule_batch(n, 64, seed=20260926, min_atoms=8)
    flat = util.flatten_molecules(mols, 64)
    fps = MorganFingerprintGenerator(2, 2048).GetFingerprintsFromInvariants(*flat, max_atoms=64).torch()
    want_fp = oracle.morgan_fingerprints(*flat, 64, 2, 2048)
    assert np.array_equal(fps.cpu().numpy().view(np.uint32), want_fp)
    sim = crossTanimotoSimilarity(fps).torch()                      # 10k x 10k float64 on the device (800 MB)
    assert sim.shape == (n, n)
    rows = np.r_[0:64, 4968:5032, n - 64:n]                          # 192 full rows against the CPU oracle, bit for bit
    assert np.array_equal(sim[rows].cpu().numpy(), oracle.cross_similarity(want_fp[rows], want_fp))
    # every entry: symmetric, in [0, 1], unit diagonal wherever the fingerprint is not empty
    assert torch.equal(sim, sim.T)
    assert float(sim.min()) >= 0.0 and float(sim.max()) <= 1.0
    nonempty = torch.from_numpy((want_fp != 0).any(1)).cuda()
    assert torch.equal(sim.diagonal() == 1.0, nonempty)
    # a checksum of row checksums against the oracle on a strided sample of columns
    cols = np.arange(0, n, 37)
    want = oracle.cross_similarity(want_fp, want_fp[cols])
    assert np.array_equal(sim[:, torch.from_numpy(cols).cuda()].cpu().numpy(), want)


def test_cfg1_on_the_reference_benchmark_smiles():
    """BASELINE.json configs[0] on the molecules the reference benchmarks with: 10 000 ChEMBL SMILES -> the library's own
    ingestion -> Morgan r = 2 / 2048 bit on the GPU -> 10k x 10k Tanimoto, against oracle Morgan -> oracle similarity."""
    from pathlib import Path

    from nvmolkit_amd.fingerprints import SmilesSet

    smiles = [line.split()[0] for line in (Path(__file__).parent / "golden" / "chembl_10k.smi").read_text().splitlines() if line.strip()]
    assert len(smiles) == 10_000
    mols = SmilesSet(smiles)
    assert np.all(mols.status == 0)
    fps = MorganFingerprintGenerator(2, 2048).GetFingerprintsFromSmiles(mols).torch()
    got = fps.cpu().numpy().view(np.uint32)
    size = np.maximum(mols.n_atoms, mols.n_bonds)
    want_fp = np.zeros_like(got)
    lo = 0
    for stride in (32, 64, 128, 256, 512, 1024):
        idx = np.flatnonzero((size >= lo) & (size < stride))
        lo = stride
        if len(idx):
            want_fp[idx] = oracle.morgan_fingerprints(*mols.morgan_inputs(idx, stride), stride, 2, 2048)
    assert np.array_equal(got, want_fp)
    density = np.unpackbits(got.view(np.uint8), axis=1).sum(1)
    assert 20 < np.median(density) < 80                              # the ~2.3 % density the synthetic sets assume
    sim = crossTanimotoSimilarity(fps).torch()
    rows = np.r_[0:64, 5000:5064, 9936:10_000]
    assert np.array_equal(sim[rows].cpu().numpy(), oracle.cross_similarity(want_fp[rows], want_fp))
    assert torch.equal(sim, sim.T) and bool((sim.diagonal() == 1.0).all())


@pytest.fixture(scope="module")
def cfg3():
    lib = synthetic.druglike_library(10_000, seed=20260926)
    molset = FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in lib])
    dev = embed_flat(molset, confs_per_molecule=10, max_iterations=10, seed=1, output=CoordinateOutput.DEVICE)
    opt = mmffOptimization.optimize_device([m["mmff"] for m in lib], dev, max_iters=200)
    torch.cuda.synchronize()
    return lib, dev, opt


def test_cfg3_etkdg_counts_and_bounds(cfg3):
    lib, dev, _ = cfg3
    n_conf = dev.num_conformers
    assert n_conf >= 0.97 * 10 * len(lib)                            # nearly every molecule gets its 10 conformers
    per_mol = np.bincount(dev.mol_indices.torch().cpu().numpy(), minlength=len(lib))
    assert per_mol.max() <= 10 and (per_mol == 10).mean() > 0.95
    # distance bounds of the embedded conformers (what the DG stage minimises): sampled molecules, every pair
    xyz = dev.values.torch().cpu().numpy()
    a_s = dev.atom_starts.torch().cpu().numpy()
    mol_of = dev.mol_indices.torch().cpu().numpy()
    rng = np.random.default_rng(0)
    worst = []
    for c in rng.choice(n_conf, size=400, replace=False):
        m = lib[mol_of[c]]
        pairs, lb, ub = m["bounds"]
        p = xyz[a_s[c]:a_s[c + 1]]
        d = np.linalg.norm(p[pairs[:, 0]] - p[pairs[:, 1]], axis=1)
        worst.append(float(np.max(np.maximum(np.maximum(lb - d, d - ub), 0.0) / ub)))
    # the stage accepts E / atom < 0.05 with E = sum (d^2 / ub^2 - 1)^2: single pairs may be off by several per cent, and in
    # the extreme ONE pair of a 96-atom molecule may carry the whole allowance: ((1 + v)^2 - 1)^2 <= 0.05 x 96 gives v <= 0.79
    # (which of the ~98 000 conformers the 400 sampled ones are depends on the batch size; 0.50 has been seen)
    assert np.median(worst) < 0.05 and np.percentile(worst, 95) < 0.12 and max(worst) < 0.79


def test_cfg3_mmff_energies_decrease_and_match_oracle_energy(cfg3):
    lib, dev, opt = cfg3
    tables = [m["mmff"] for m in lib]
    a_s = dev.atom_starts.torch().cpu().numpy()
    mol_of = dev.mol_indices.torch().to(torch.int32)
    batch = FlatForcefieldBatch(MMFF, a_s, stack_molecule_tables(MMFF, tables), system_mol=mol_of)
    e0 = batch.compute_energy(dev.values.torch().reshape(-1).contiguous())
    e1 = opt.energies.torch()
    assert bool((e1 <= e0 + 1e-9).all())
    assert torch.allclose(batch.compute_energy(opt.values.torch().reshape(-1).contiguous()), e1, rtol=1e-9, atol=1e-9)
    assert float((e1 / torch.from_numpy(np.diff(a_s)).cuda()).median()) < 5.0      # kcal/mol per atom: relaxed structures
    # reported energies against the C oracle on a sample of the optimised conformers
    sel = np.random.default_rng(1).choice(dev.num_conformers, size=256, replace=False)
    sel.sort()
    xyz = opt.values.torch().cpu().numpy()
    sizes = np.diff(a_s)[sel]
    sub_as = np.concatenate([[0], np.cumsum(sizes)])
    sub_pos = np.concatenate([xyz[a_s[c]:a_s[c + 1]].reshape(-1) for c in sel])
    cpu = ffc.Batch(MMFF, sub_as, stack_molecule_tables(MMFF, tables), system_mol=mol_of.cpu().numpy()[sel])
    np.testing.assert_allclose(e1.cpu().numpy()[sel], cpu.energy(sub_pos), rtol=1e-9, atol=1e-8)
    conv = opt.converged.torch().float().mean().item()
    assert 0.0 <= conv <= 1.0                                        # reported by bench.py; BFGS from H = I needs ~2.4 n iterations


def test_mmff_minima_exist_and_are_reached_given_enough_iterations():
    """At the benchmark's maxIters = 200 only a few per cent of 48-atom conformers meet the gradient tolerance (BFGS from
    H = I needs about 2.4 n iterations); the same conformers converge when they are given the iterations, and the C oracle
    (same algorithm on the CPU) agrees on both fractions."""
    lib = synthetic.druglike_library(96, seed=11, processes=1)
    tables = [m["mmff"] for m in lib]
    dev = embed_flat(FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in lib]), confs_per_molecule=2, max_iterations=10, seed=5,
                     output=CoordinateOutput.DEVICE)
    short = mmffOptimization.optimize_device(tables, dev, max_iters=200)
    long_ = mmffOptimization.optimize_device(tables, dev, max_iters=2000)
    f_short = short.converged.torch().float().mean().item()
    f_long = long_.converged.torch().float().mean().item()
    assert f_long >= 0.9 and f_short < f_long
    assert bool((long_.energies.torch() <= short.energies.torch() + 1e-6).all())
    # the oracle on the same start coordinates
    a_s = dev.atom_starts.torch().cpu().numpy()
    mol_of = dev.mol_indices.torch().cpu().numpy().astype(np.int32)
    cpu = ffc.Batch(MMFF, a_s, stack_molecule_tables(MMFF, tables), system_mol=mol_of)
    pos = dev.values.torch().cpu().numpy().reshape(-1).copy()
    _, _, st200, _ = cpu.minimize(pos, max_iters=200)
    _, _, st2000, _ = cpu.minimize(pos, max_iters=2000)
    assert abs(float(np.mean(st200 == 0)) - f_short) < 0.15
    assert float(np.mean(st2000 == 0)) >= 0.9


def test_etkdg_is_bitwise_reproducible_for_a_seed():
    lib = synthetic.druglike_library(64, seed=5, processes=1)
    molset = FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in lib])
    a = embed_flat(molset, confs_per_molecule=4, max_iterations=10, seed=7)
    b = embed_flat(molset, confs_per_molecule=4, max_iterations=10, seed=7)
    assert np.array_equal(a.conf_counts, b.conf_counts) and np.array_equal(a.stage_failures, b.stage_failures)
    assert torch.equal(a.coords, b.coords)
    c = embed_flat(molset, confs_per_molecule=4, max_iterations=10, seed=8)
    assert not torch.equal(a.coords, c.coords)


def test_etkdg_pipeline_matches_oracle_pipeline_statistically():
    """Same molecules, same seed, same scheduler and start coordinates on both sides (the C oracle restates the whole
    stage pipeline): conformer counts, per-stage failure totals and the geometry of the conformers agree.  Individual
    trajectories of 400-iteration minimisations are chaotic in the last digits, so agreement is per population."""
    lib = synthetic.druglike_library(96, seed=9, processes=1)
    mols = [FlatMolecule(**m["embed"]) for m in lib]
    gpu = embed_flat(FlatMoleculeSet(mols), confs_per_molecule=4, max_iterations=10, seed=3)
    coords, counts, slots, fails, _ = ffc.etkdg_embed(mols, confs_per_molecule=4, max_iterations=10, seed=3, batch_size=4096)
    # (tools/etkdg_population_parity.py on 1500 molecules: 5860 conformers on both sides, every molecule with the same count,
    # stage failures [0 23 458 0 0 0 1 0 5 0 0] against [0 23 459 0 0 0 1 0 6 0 0], Kolmogorov-Smirnov distance of the
    # violation distributions 0.009 — profiles/r03_conformers/etkdg_population_parity.json)
    assert abs(int(gpu.conf_counts.sum()) - int(counts.sum())) <= 0.01 * counts.sum()
    assert np.mean(np.asarray(gpu.conf_counts) == np.asarray(counts)) >= 0.97
    assert np.all(np.abs(gpu.stage_failures - fails) <= np.maximum(2, 0.15 * np.maximum(gpu.stage_failures, fails)))
    # Geometry: both sides start every attempt from the same coordinates, but 200-400 iteration minimisations on a
    # multi-minimum landscape amplify last-digit differences into different (equally valid) embeddings — measured: the
    # inter-atomic distances of the first conformers agree to 1e-2 A for 1 molecule in 95.  So the populations are compared:
    # every conformer of either side satisfies the distance bounds the same way.
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
    assert np.median(vg) < 0.05 and np.median(vc) < 0.05 and vg.max() < 0.3 and vc.max() < 0.3
    assert abs(np.median(vg) - np.median(vc)) < 0.02 and abs(np.percentile(vg, 90) - np.percentile(vc, 90)) < 0.04
