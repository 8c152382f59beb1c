"""Aromaticity perception of the SMILES ingestion (SmilesSet(..., perceive_aromaticity=True)) against RDKit itself, without
RDKit: the reference's ChEMBL SMILES were written by RDKit in aromatic form, so their lower-case atoms ARE RDKit's perception.
Every aromatic molecule is turned into a Kekule form by the oracle, written out with bracket atoms and handed to the library,
which must give back exactly the aromatic atoms and bonds RDKit recorded — for every one of them, the 15 porphyrins (two C=C
stay double: the inner 16-ring shares two bonds with each pyrrole ring, so RDKit does not treat them as fused) and the 7
fullerene adducts (atoms shared by three rings) included."""

from pathlib import Path

import numpy as np
import pytest

from nvmolkit_amd.fingerprints import SmilesSet
from oracle import aromaticity as oarom
from oracle import smiles as osmi

GOLDEN = Path(__file__).parent / "golden"


def _lines(name):
    return [line.split()[0] for line in (GOLDEN / name).read_text().splitlines() if line.strip() and not line.startswith("#")]


def _kekule_forms(smiles):
    """[(index, kekule bracket SMILES, atom order of that string, reference atom table, reference bond table)] of the
    aromatic molecules of `smiles` (tables from the oracle's parse of the aromatic form)."""
    out = []
    for i, smi in enumerate(smiles):
        atoms, bonds = osmi.molecule(smi)
        if not (bonds[:, 2] == 12).any():
            continue
        kek = oarom.kekulize(atoms, bonds)
        text, order = oarom.write_bracket_smiles(atoms, kek, bonds)
        out.append((i, text, order, atoms, bonds))
    return out


def _check(smiles, max_refused):
    forms = _kekule_forms(smiles)
    got = SmilesSet([f[1] for f in forms], perceive_aromaticity=True)
    refused = SmilesSet([f[1] for f in forms], perceive_aromaticity=False)   # strict mode: Kekule-form aromatic rings are refused
    assert np.all(refused.status == 3)
    n_refused = 0
    for j, (i, text, order, atoms, bonds) in enumerate(forms):
        if got.status[j] == 3:                                       # a fused system with too many ring combinations
            n_refused += 1
            continue
        assert got.status[j] == 0, (smiles[i], int(got.status[j]))
        ga, gb = got.graph(j)
        inv = np.empty(len(order), dtype=np.int64)                   # original atom -> position in the written string
        inv[np.asarray(order)] = np.arange(len(order))
        assert np.array_equal(ga[inv][:, [0, 1, 2, 3, 5]], atoms[:, [0, 1, 2, 3, 5]]), smiles[i]
        assert np.array_equal(ga[inv][:, 4], atoms[:, 4]), smiles[i]                       # the aromatic atoms RDKit wrote
        want = {(min(inv[a], inv[b]), max(inv[a], inv[b])): t for a, b, t, _ in bonds}
        have = {(min(a, b), max(a, b)): t for a, b, t, _ in gb}
        assert have == want, smiles[i]                                                     # and the aromatic bonds
    assert n_refused <= max_refused
    return len(forms), n_refused


def test_perception_reproduces_rdkits_aromaticity_on_chembl_1k():
    n, refused = _check(_lines("chembl_1k.smi"), max_refused=0)
    assert n > 700


def test_perception_reproduces_rdkits_aromaticity_on_chembl_10k():
    n, refused = _check(_lines("chembl_10k.smi"), max_refused=0)     # 8864 aromatic molecules, none refused
    assert n > 8000 and refused == 0


def test_perception_reproduces_rdkits_aromaticity_on_the_other_reference_smiles():
    """2 300 more SMILES written by RDKit: the peptides of nvmolkit/tests/testdata/smiles.csv, butina_test_smiles.csv and
    benchmarks/data/benchmark_smiles.csv.  Strict mode accepts all of them as written, and their Kekule forms come back."""
    smiles = _lines("more_rdkit_smiles.smi")
    assert len(smiles) == 2300 and np.all(SmilesSet(smiles, perceive_aromaticity=False).status == 0)
    n, refused = _check(smiles, max_refused=0)
    assert n > 700                                                   # the aromatic ones among them (the benchmark file is mostly aliphatic)


def test_oracle_perception_agrees_with_the_library_on_kekule_forms():
    forms = _kekule_forms(_lines("chembl_1k.smi")[:300])
    got = SmilesSet([f[1] for f in forms], perceive_aromaticity=True)
    for j, (i, text, order, atoms, bonds) in enumerate(forms):
        if got.status[j] != 0:
            continue
        ka, kb = osmi.molecule(text)                                 # the oracle's own parse of the Kekule string
        arom, types = oarom.perceive(ka, kb)
        ga, gb = got.graph(j)
        assert np.array_equal(ga[:, 4].astype(bool), arom) and np.array_equal(gb[:, 2], types), text


@pytest.mark.parametrize("smi,n_aromatic_bonds", [
    ("C1=CC=CC=C1", 6), ("C1=CC=NC=C1", 6), ("C1=CNC=C1", 5), ("C1=COC=C1", 5), ("C1=CSC=C1", 5), ("C1=CN=CN1", 5),
    ("C1=CC=C2C=CC=CC2=C1", 11), ("C1=CC=C2NC=CC2=C1", 10), ("C1=CC2=CC=CC=CC2=C1", 10),      # azulene: the fusion bond stays single
    ("O=C1C=CC=CN1", 6), ("O=C1C=COC=C1", 6), ("O=C1C=CC=CC=C1", 7), ("[CH-]1C=CC=C1", 5), ("[CH+]1C=CC=CC=C1", 7),
    ("CN1C=NC2=C1C(=O)N(C)C(=O)N2C", 10),                                                       # caffeine: both rings
    ("C=C1C=CC=CC1=C", 6),                                                                      # o-xylylene: exocyclic C=C gives 1 electron each
    ("C1=CCCCC1", 0), ("O=C1C=CC(=O)C=C1", 0), ("C1=CC=CC1", 0), ("C=C1C=CC=C1", 0), ("C1=CC2=CC=CC2=C1", 0),
    ("C1=CC=CC=CC=C1", 0), ("C1=CC=CCC=C1", 0), ("O=C1C=CC=C1", 0), ("N1=P(Cl)(Cl)N=P(Cl)(Cl)N=P1(Cl)Cl", 0),
    ("C1=CC2=CC=C3C=CC=C4C=CC(=C1)C2=C34", 19),                                                 # pyrene: the two inner atoms lie in three rings
    ("C1=CC2=CC=C3C=CC4=CC=C5C=CC6=CC=C1C7=C6C5=C4C3=C27", 30),                                # coronene
    ("C1=CC2=CC3=CC=C(N3)C=C4C=CC(=N4)C=C5C=CC(=CC1=N2)N5", 22),                               # porphine: two C=C stay double (RDKit)
    ("C1=CC=CC=CC=CC=CC=CC=CC=CC=C1", 18)])                                                     # [18]annulene: a single 18-ring
def test_textbook_rings(smi, n_aromatic_bonds):
    s = SmilesSet([smi], perceive_aromaticity=True)
    assert s.status[0] == 0
    _, bonds = s.graph(0)
    assert int((bonds[:, 2] == 12).sum()) == n_aromatic_bonds
    assert int(SmilesSet([smi], perceive_aromaticity=False).status[0]) == (3 if n_aromatic_bonds else 0)      # strict mode: refused iff aromatic


def test_kekule_and_aromatic_forms_give_the_same_morgan_inputs():
    pairs = [("c1ccccc1O", "C1=CC=CC=C1O"), ("c1ccc2[nH]ccc2c1", "C1=CC=C2NC=CC2=C1"), ("Cn1cnc2c1c(=O)n(C)c(=O)n2C", "CN1C=NC2=C1C(=O)N(C)C(=O)N2C"),
             ("O=c1cccc[nH]1", "O=C1C=CC=CN1"), ("CC(=O)Nc1ccc(O)cc1", "CC(=O)NC1=CC=C(O)C=C1")]
    for arom, kek in pairs:
        a = SmilesSet([arom]).morgan_inputs([0], 32)
        k = SmilesSet([kek], perceive_aromaticity=True).morgan_inputs([0], 32)
        for x, y in zip(a, k):
            assert np.array_equal(x, y), (arom, kek)


def test_rdkit_written_smiles_survive_kekulisation_and_perception_unchanged():
    """What the strict mode does with every molecule: Kekulise the aromatic form, perceive again, compare with what was
    written.  All 10 000 SMILES of the reference's benchmark file were written by RDKit, so none may be refused and applying
    the perceived aromaticity (perceive_aromaticity=True) must give the very same graphs."""
    smiles = _lines("chembl_10k.smi")
    default, applied = SmilesSet(smiles, perceive_aromaticity=False), SmilesSet(smiles, perceive_aromaticity=True)
    assert np.all(default.status == 0) and np.all(applied.status == 0)
    for i in range(len(smiles)):
        (a0, b0), (a1, b1) = default.graph(i), applied.graph(i)
        assert np.array_equal(a0, a1) and np.array_equal(b0, b1), smiles[i]


@pytest.mark.parametrize("written,perceived", [
    ("c1ccccccc1", "C1=CC=CC=CC=C1"),              # cyclooctatetraene is not aromatic however it is written
    ("C1=CC=CC=C1", "c1ccccc1"), ("c1ccc2ccccc2c1", "C1=CC=C2C=CC=CC2=C1"),
    ("O=c1cc[nH]cc1", "O=C1C=CNC=C1")])
def test_applied_perception_does_not_depend_on_how_the_input_was_written(written, perceived):
    a, b = SmilesSet([written], perceive_aromaticity=True), SmilesSet([perceived], perceive_aromaticity=True)
    assert a.status[0] == 0 and b.status[0] == 0
    for x, y in zip(a.graph(0), b.graph(0)):
        assert np.array_equal(x, y)


def test_rdkit_book_examples_of_fused_systems():
    """Two molecules the RDKit Book's "Aromaticity" section spells out.  'O=C1C=CC(=O)C2=C1OC=CO2': atoms 6 and 7 are aromatic
    and the bond between them is not (m.GetAtomWithIdx(6).GetIsAromatic() -> True, ...(7)... -> True,
    m.GetBondBetweenAtoms(6,7).GetIsAromatic() -> False) — neither ring is aromatic by itself, their envelope is.
    Biphenylene 'C1=CC2=C(C=C1)C1=CC=CC=C21' is written back as 'c1ccc2c(c1)-c1ccccc1-2': two benzene rings, the four-ring's
    other two bonds single."""
    s = SmilesSet(["O=C1C=CC(=O)C2=C1OC=CO2", "C1=CC2=C(C=C1)C1=CC=CC=C21"])
    assert np.all(s.status == 0)
    atoms, bonds = s.graph(0)
    assert atoms[6, 4] == 1 and atoms[7, 4] == 1 and int(atoms[:, 4].sum()) == 10
    assert [int(t) for a, b, t, _ in bonds if {int(a), int(b)} == {6, 7}] == [2] and int((bonds[:, 2] == 12).sum()) == 10
    atoms, bonds = s.graph(1)
    assert int(atoms[:, 4].sum()) == 12 and int((bonds[:, 2] == 12).sum()) == 12
    assert sorted((int(a), int(b), int(t)) for a, b, t, _ in bonds if t != 12) == [(2, 11, 1), (3, 6, 1)]
