"""Force-field and ETKDG parity with RDKit — SKIPPED wherever RDKit is not installed, which includes both images this project
is built and tested in: these tests have never run.  They restate the reference's own acceptance criteria for the conformer
half of the hot path, so that the first host with RDKit + an MI355X turns DESIGN.md's "parity unpinned" rows into measured ones:

* per-term MMFF energies 5e-5 / gradients 1e-4 against RDKit's own terms (reference: tests/test_mmff.cu:53-55, :795-1016),
  through the flattener a caller of ``MMFFOptimizeMoleculesConfs`` goes through (``flatten_mmff_from_rdkit``);
* the same for UFF through ``flatten_uff_from_rdkit`` (reference: tests/test_uff.cu);
* the minimised MMFF energies the reference pins: 26.8743 / 66.1801 / -18.7326 / -207.436 for the first four molecules of
  MMFF94_dative.sdf from perturbed starts, 33.0842 for 50_atom_mol.sdf (tests/test_mmff.cu:1521, :1555, :1608; tolerance 1e-3);
* the reference's ETKDG acceptance (nvmolkit/tests/test_embed_molecules.py:115-260): as many conformers as RDKit
  makes, and at least half of them within 0.2 A RMSD of one of RDKit's, for every ETKDG variant;
* the distance-geometry field against RDKit's own (rdDistGeom.GetMoleculeBoundsMatrix -> first-term bounds of
  tests/test_flattened_builder.cu:188-189)."""

from pathlib import Path

import numpy as np
import pytest

Chem = pytest.importorskip("rdkit.Chem", reason="RDKit is not installed")
pytestmark = pytest.mark.gpu

import torch  # noqa: E402
from rdkit.Chem import AllChem, rdDistGeom, rdForceFieldHelpers  # noqa: E402

from nvmolkit_amd import embedMolecules, mmffOptimization, uffOptimization  # noqa: E402
from nvmolkit_amd.forcefield import MMFF, UFF, FlatForcefieldBatch  # noqa: E402
from nvmolkit_amd.types import HardwareOptions  # noqa: E402

GOLDEN = Path(__file__).parent / "golden"
FUNCTION_E_TOL, GRAD_TOL, MINIMIZE_E_TOL = 5.0e-5, 1.0e-4, 1.0e-3   # tests/test_mmff.cu:53-55
MMFF_TERMS = ["Bond", "Angle", "StretchBend", "Oop", "Torsion", "VdW", "Ele"]   # term group g of the MMFF layout


def molecules(name, limit=None):
    mols = [m for m in Chem.SDMolSupplier(str(GOLDEN / name), removeHs=False, sanitize=True) if m is not None]
    return mols[:limit] if limit else mols


def perturbed(mol, delta, seed):
    xyz = mol.GetConformer().GetPositions()
    return xyz + np.random.default_rng(seed).uniform(-delta, delta, xyz.shape)


def rdkit_mmff(mol, only=None):
    props = rdForceFieldHelpers.MMFFGetMoleculeProperties(mol)
    if only is not None:
        for term in MMFF_TERMS:
            getattr(props, f"SetMMFF{term}Term")(term == only)
    return props, rdForceFieldHelpers.MMFFGetMoleculeForceField(mol, props)


def batch_of(kind, groups, n_atoms, only=None):
    if only is not None:
        groups = [g if k == only else (g[0][:0], g[1][:0]) for k, g in enumerate(groups)]
    stacked = [(np.array([0, len(idx)], dtype=np.int32), idx, par) for idx, par in groups]
    return FlatForcefieldBatch(kind, np.array([0, n_atoms], dtype=np.int32), stacked)


@pytest.mark.parametrize("name", ["MMFF94_dative_every4th.sdf", "MMFF94_hypervalent_every4th.sdf", "larger_molecules.sdf"])
def test_mmff_terms_equal_rdkit(name):
    """Every MMFF term kind on its own, then all together: energy and gradient of the kernels against RDKit's force field
    (tests/test_mmff.cu:795-1016 does this per term against RDKit's contribs)."""
    for m, mol in enumerate(molecules(name)):
        if not rdForceFieldHelpers.MMFFHasAllMoleculeParams(mol):
            continue
        props_all, _ = rdkit_mmff(mol)
        groups = mmffOptimization.flatten_mmff_from_rdkit(mol, props_all)
        pos = torch.from_numpy(mol.GetConformer().GetPositions().reshape(-1).copy()).cuda()
        for k, term in list(enumerate(MMFF_TERMS)) + [(None, None)]:
            _, ff = rdkit_mmff(mol, term)
            gpu = batch_of(MMFF, groups, mol.GetNumAtoms(), k)
            e = float(gpu.compute_energy(pos)[0])
            g = gpu.compute_gradient(pos).cpu().numpy()
            assert abs(e - ff.CalcEnergy()) <= FUNCTION_E_TOL, (name, m, term, e, ff.CalcEnergy())
            assert np.max(np.abs(g - np.array(ff.CalcGrad()))) <= GRAD_TOL, (name, m, term)


@pytest.mark.parametrize("name", ["MMFF94_dative_every4th.sdf", "larger_molecules.sdf"])
def test_uff_energy_and_gradient_equal_rdkit(name):
    for m, mol in enumerate(molecules(name)):
        if not rdForceFieldHelpers.UFFHasAllMoleculeParams(mol):
            continue
        try:
            groups = uffOptimization.flatten_uff_from_rdkit(mol)
        except NotImplementedError:   # sp2 centres in 3- / 4-rings, 5-coordinate centres: documented refusals of the flattener
            continue
        ff = rdForceFieldHelpers.UFFGetMoleculeForceField(mol)
        gpu = batch_of(UFF, groups, mol.GetNumAtoms())
        pos = torch.from_numpy(mol.GetConformer().GetPositions().reshape(-1).copy()).cuda()
        assert abs(float(gpu.compute_energy(pos)[0]) - ff.CalcEnergy()) <= FUNCTION_E_TOL, (name, m)
        assert np.max(np.abs(gpu.compute_gradient(pos).cpu().numpy() - np.array(ff.CalcGrad()))) <= GRAD_TOL, (name, m)


def test_mmff_minimised_energies_are_the_references_known_answers():
    """tests/test_mmff.cu:1521-1608: ten perturbed conformers (0.5 A) of each of the first four molecules of MMFF94_dative.sdf all
    minimise to 26.8743 / 66.1801 / -18.7326 / -207.436; the reported energy equals RDKit's at the returned coordinates."""
    want = [26.8743, 66.1801, -18.7326, -207.436]
    mols = molecules("MMFF94_dative_first5.sdf", 4)
    for mol in mols:
        base = Chem.Conformer(mol.GetConformer())
        for c in range(1, 10):
            conf = Chem.Conformer(base)
            for i, p in enumerate(perturbed(mol, 0.5, c + mol.GetNumAtoms())):
                conf.SetAtomPosition(i, p.tolist())
            mol.AddConformer(conf, assignId=True)
    got = mmffOptimization.MMFFOptimizeMoleculesConfs(mols, maxIters=1000)
    for m, (mol, energies) in enumerate(zip(mols, got)):
        _, ff = rdkit_mmff(mol)
        assert len(energies) == mol.GetNumConformers() == 10
        for conf, e in zip(mol.GetConformers(), energies):
            at = ff.CalcEnergy(conf.GetPositions().reshape(-1).tolist())
            assert abs(e - at) <= MINIMIZE_E_TOL, (m, e, at)
            assert abs(at - want[m]) <= MINIMIZE_E_TOL, (m, at, want[m])


def test_mmff_minimised_energy_of_the_50_atom_molecule():
    """tests/test_mmff.cu:1597-1620: 50_atom_mol.sdf perturbed by 0.5 A minimises to 33.0842."""
    mol = next(m for m in molecules("larger_molecules.sdf") if m.GetNumAtoms() == 50)
    conf = mol.GetConformer()
    for i, p in enumerate(perturbed(mol, 0.5, 0)):
        conf.SetAtomPosition(i, p.tolist())
    energies = mmffOptimization.MMFFOptimizeMoleculesConfs([mol], maxIters=1000)[0]
    _, ff = rdkit_mmff(mol)
    at = ff.CalcEnergy(mol.GetConformer().GetPositions().reshape(-1).tolist())
    assert abs(energies[0] - at) <= MINIMIZE_E_TOL and abs(at - 33.0842) <= MINIMIZE_E_TOL


def test_uff_optimisation_reaches_rdkit_minima():
    """nvmolkit/tests/test_uff_optimization.py: energies after UFFOptimizeMoleculesConfs equal RDKit's own optimisation of the
    same starts (1e-3 relative, same local minimum from a 0.1 A perturbation)."""
    mols = molecules("MMFF94_dative_first5.sdf")
    ref = [Chem.Mol(m) for m in mols]
    got = uffOptimization.UFFOptimizeMoleculesConfs(mols, maxIters=1000)
    for m, (mol, r, energies) in enumerate(zip(mols, ref, got)):
        res = rdForceFieldHelpers.UFFOptimizeMoleculeConfs(r, maxIters=1000)
        assert abs(energies[0] - res[0][1]) <= 1e-3 * max(1.0, abs(res[0][1])), (m, energies[0], res[0][1])


ETKDG_VARIANTS = {"ETKDG": rdDistGeom.ETKDG, "ETKDGv2": rdDistGeom.ETKDGv2, "ETKDGv3": rdDistGeom.ETKDGv3,
                  "srETKDGv3": rdDistGeom.srETKDGv3, "KDG": rdDistGeom.KDG, "ETDG": rdDistGeom.ETDG, "DG": rdDistGeom.KDG}


@pytest.mark.parametrize("variant", list(ETKDG_VARIANTS))
@pytest.mark.parametrize("batched", [False, True])
def test_embed_molecules_meets_the_references_acceptance(variant, batched):
    """nvmolkit/tests/test_embed_molecules.py:190-330: same conformer counts as rdDistGeom.EmbedMultipleConfs, and at least half of
    our conformers within 0.2 A RMSD (heavy + hydrogen atoms, aligned) of one of RDKit's."""
    confs = 5
    params = ETKDG_VARIANTS[variant]()
    if variant == "DG":
        params.useBasicKnowledge = False
    params.useRandomCoords = True
    params.randomSeed = 42
    base = molecules("MMFF94_dative_first5.sdf")
    for m in base:
        m.RemoveAllConformers()
    ours, theirs = [Chem.Mol(m) for m in base], [Chem.Mol(m) for m in base]
    for mol in theirs:
        rdDistGeom.EmbedMultipleConfs(mol, numConfs=confs, params=params)
    if batched:
        embedMolecules.EmbedMolecules(ours, params, confsPerMolecule=confs, maxIterations=-1)
    else:
        for mol in ours:
            embedMolecules.EmbedMolecules([mol], params, confsPerMolecule=confs, maxIterations=-1,
                                          hardwareOptions=HardwareOptions(preprocessingThreads=1, batchSize=1, batchesPerGpu=1))
    for m, (a, b) in enumerate(zip(ours, theirs)):
        assert a.GetNumConformers() == b.GetNumConformers() == confs, (variant, m)
        both = Chem.Mol(b)
        for conf in a.GetConformers():
            both.AddConformer(conf, assignId=True)
        n_ref = b.GetNumConformers()
        similar = 0
        for k in range(n_ref, n_ref + a.GetNumConformers()):
            best = min(AllChem.GetConformerRMS(both, r, k, prealigned=False) for r in range(n_ref))
            similar += best <= 0.2
        assert similar >= 0.5 * a.GetNumConformers(), (variant, m, similar)


def test_distance_geometry_terms_come_from_rdkits_bounds_matrix():
    """tests/test_flattened_builder.cu:148-189: the first distance term of the first molecule carries lb^2 / ub^2 of RDKit's
    smoothed bounds matrix; the DG energy of RDKit's own embedding is (nearly) zero in our field."""
    from nvmolkit_amd import _rdkit_embed

    mol = molecules("MMFF94_dative_first5.sdf", 1)[0]
    params = rdDistGeom.ETKDGv3()
    params.useRandomCoords = True
    flat = _rdkit_embed.flatten_etkdg_from_rdkit(mol, params)
    bm = rdDistGeom.GetMoleculeBoundsMatrix(mol)
    idx, par = flat["dg"][0]
    i, j = int(idx[0, 0]), int(idx[0, 1])
    assert np.isclose(par[0, 0], bm[max(i, j), min(i, j)] ** 2, rtol=1e-5) and np.isclose(par[0, 1], bm[min(i, j), max(i, j)] ** 2, rtol=1e-5)
