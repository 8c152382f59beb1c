"""Direct comparison with RDKit — SKIPPED wherever RDKit is not installed, which includes both images this project is built
and tested in: these tests have never run there.  They are the check a maintainer with RDKit can run to replace the
indirect pins of DESIGN.md section 2 (aromaticity and hydrogen counts recovered from RDKit-written files) by the real
thing: the graphs of the RDKit-free ingestion against Chem.MolFromSmiles / SDMolSupplier, the Morgan invariants against
RDKit's atom-invariant generator and — on a GPU — the fingerprints against rdFingerprintGenerator."""

from pathlib import Path

import numpy as np
import pytest

Chem = pytest.importorskip("rdkit.Chem", reason="RDKit is not installed")

from nvmolkit_amd.fingerprints import MorganFingerprintGenerator, SmilesSet  # noqa: E402

GOLDEN = Path(__file__).parent / "golden"


def rdkit_tables(mol):
    ring = mol.GetRingInfo()
    atoms = np.array([[a.GetAtomicNum(), a.GetFormalCharge(), a.GetIsotope(), a.GetTotalNumHs(), int(a.GetIsAromatic()),
                       int(ring.NumAtomRings(a.GetIdx()) > 0)] for a in mol.GetAtoms()], dtype=np.int64).reshape(-1, 6)
    bonds = {(min(b.GetBeginAtomIdx(), b.GetEndAtomIdx()), max(b.GetBeginAtomIdx(), b.GetEndAtomIdx())):
             (int(b.GetBondType()), int(ring.NumBondRings(b.GetIdx()) > 0)) for b in mol.GetBonds()}
    return atoms, bonds


def assert_same_graph(got_atoms, got_bonds, mol, label):
    atoms, bonds = rdkit_tables(mol)
    assert np.array_equal(got_atoms, atoms), label
    assert {(min(a, b), max(a, b)): (t, r) for a, b, t, r in got_bonds.tolist()} == bonds, label


def smiles_of(name):
    return [line.split()[0] for line in (GOLDEN / name).read_text().splitlines() if line.strip() and not line.startswith("#")]


@pytest.mark.parametrize("name", ["chembl_1k.smi", "more_rdkit_smiles.smi", "chembl_10k.smi"])
def test_smiles_graphs_equal_rdkit(name):
    smiles = smiles_of(name)
    got = SmilesSet(smiles)
    for i, smi in enumerate(smiles):
        mol = Chem.MolFromSmiles(smi)
        assert (mol is None) == (got.status[i] != 0), smi
        if mol is not None:
            assert_same_graph(*got.graph(i), mol, smi)


@pytest.mark.parametrize("smi", ["C1=CC=CC=C1", "CN(=O)=O", "C1=CC=CN(=O)=C1", "CN=N#N", "O=Cl(=O)O", "c1ccccccc1", "[H]P([H])([H])=O",
                                 "C1=CC2=CC3=CC=C(N3)C=C4C=CC(=N4)C=C5C=CC(=CC1=N2)N5", "C1=CC2=CC=C3C=CC=C4C=CC(=C1)C2=C34",
                                 "c1cccc1", "c1ccnc1", "C(C)(C)(C)(C)C", "C[N](C)(C)C", "[CH3]", "[13CH4]", "[2H]O[2H]", "[Na+].[Cl-]"])
def test_sanitisation_cases_equal_rdkit(smi):
    mol = Chem.MolFromSmiles(smi)
    got = SmilesSet([smi])
    assert (mol is None) == (got.status[0] != 0), smi
    if mol is not None:
        assert_same_graph(*got.graph(0), mol, smi)


@pytest.mark.parametrize("name", ["MMFF94_dative_every4th.sdf", "MMFF94_hypervalent_every4th.sdf", "larger_molecules.sdf"])
def test_sd_file_graphs_equal_rdkit(name):
    got = SmilesSet.from_sdf_file(GOLDEN / name)
    for i, mol in enumerate(Chem.SDMolSupplier(str(GOLDEN / name))):
        assert (mol is None) == (got.status[i] != 0), i
        if mol is not None:
            assert_same_graph(*got.graph(i), mol, f"{name} record {i}")


def test_morgan_invariants_equal_rdkit():
    from rdkit.Chem import rdFingerprintGenerator

    smiles = smiles_of("chembl_1k.smi")
    got = SmilesSet(smiles)
    inv_gen = rdFingerprintGenerator.GetMorganAtomInvGen(includeRingMembership=True)
    ids = np.flatnonzero((got.status == 0) & (np.maximum(got.n_atoms, got.n_bonds) < 128))
    atom_inv = got.morgan_inputs(ids, 128)[0]
    for row, i in zip(atom_inv, ids):
        want = np.array(inv_gen.GetAtomInvariants(Chem.MolFromSmiles(smiles[i])), dtype=np.uint32)
        assert np.array_equal(row[:len(want)], want), smiles[i]


@pytest.mark.gpu
@pytest.mark.parametrize("radius,fp_size", [(2, 2048), (3, 1024), (0, 512)])
def test_fingerprints_equal_rdkit(radius, fp_size):
    from rdkit.Chem import rdFingerprintGenerator

    smiles = smiles_of("chembl_1k.smi")
    fps = MorganFingerprintGenerator(radius=radius, fpSize=fp_size).GetFingerprints(smiles).torch().cpu().numpy().view(np.uint32)
    gen = rdFingerprintGenerator.GetMorganGenerator(radius=radius, fpSize=fp_size)
    for row, smi in zip(fps, smiles):
        want = np.zeros(fp_size // 32, dtype=np.uint32)
        for bit in gen.GetFingerprint(Chem.MolFromSmiles(smi)).GetOnBits():
            want[bit // 32] |= np.uint32(1) << np.uint32(bit % 32)
        assert np.array_equal(row, want), smi
