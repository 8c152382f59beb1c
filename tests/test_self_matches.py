"""nvmk_smiles_self_matches — the molecule's self matches for symmetry-aware RMS pruning (reference: getMolSelfMatches,
rdkit_extensions/conformer_pruning.cpp:24-60, which asks RDKit's SubstructMatch(mol, mol, uniquify=False, maxMatches=1000) on the
hydrogen-stripped molecule).  Pinned two ways: the automorphism-group orders of molecules whose symmetry is textbook, and the
exhaustive search of oracle/smiles.py on the same graphs."""

from pathlib import Path

import numpy as np
import pytest

from nvmolkit_amd.fingerprints import SmilesSet
from oracle import smiles as osm

ROOT = Path(__file__).resolve().parents[1]

KNOWN_ORDERS = [
    ("c1ccccc1", 12),            # D6h acting on six atoms
    ("Cc1ccccc1", 2),
    ("C1CCCCC1", 12),
    ("CC(C)(C)C", 24),           # the four methyls in any order
    ("Cc1ccc(C)cc1", 4),
    ("c1ccc(cc1)-c1ccccc1", 8),
    ("CCO", 1),
    ("CC", 2),
    ("C", 1),
    ("C1CC1", 6),
    ("FC(F)(F)c1ccccc1", 12),    # 3! for the fluorines x 2 for the ring flip
    ("C1CCC2CCCCC2C1", 4),       # decalin's graph
    ("[13CH3]C", 1),             # the isotope tells the two carbons apart
    ("C[N+](C)(C)C", 24),
    ("O=C=O", 2),
    ("N#N", 2),
]


@pytest.mark.parametrize("smiles,order", KNOWN_ORDERS)
def test_group_orders_of_textbook_molecules(smiles, order):
    s = SmilesSet([smiles], perceive_aromaticity=True)
    m = s.self_matches(0)
    assert m.shape == (order, int(s.n_atoms[0]))
    assert (m[0] == np.arange(m.shape[1])).all(), "the identity comes first: the reference takes its reference atoms from match 0"
    assert len({tuple(r) for r in m}) == order
    for r in m:
        assert sorted(r) == list(range(m.shape[1]))


@pytest.mark.parametrize("smiles,with_sym,without", [("CC(=O)[O-]", 2, 1), ("CC(=O)O", 2, 1), ("C[N+](=O)[O-]", 2, 1), ("NC(=N)c1ccccc1", 4, 2),
                                                     ("CC(=O)OC", 1, 1), ("CC(=O)N", 1, 1), ("OC(=O)CC(=O)O", 8, 2),
                                                     ("NC(=N)N", 6, 2), ("NC(=O)CC(=N)O", 2, 1), ("NC(=O)CC(N)O", 1, 1)])
def test_conjugated_terminal_groups_are_interchangeable_only_on_request(smiles, with_sym, without):
    s = SmilesSet([smiles], perceive_aromaticity=True)
    assert len(s.self_matches(0, symmetrize_terminal_groups=True)) == with_sym
    assert len(s.self_matches(0, symmetrize_terminal_groups=False)) == without


@pytest.mark.parametrize("smiles", ["c1ccccc1", "CC(C)(C)C", "CC(=O)[O-]", "OC(=O)CC(=O)O", "C1CC1C", "ClC(Cl)Cl", "CC(C)C(C)C", "c1ccncc1",
                                    "C[N+](=O)[O-]", "CC.CC", "NC(=N)N", "NC(=O)CC(=N)O", "NC(=O)CC(N)O", "C1CCC1", "[2H]C([2H])O", "FC(F)=C(F)F", "O=S(=O)(C)C"])
@pytest.mark.parametrize("sym", [True, False])
def test_the_same_set_as_an_exhaustive_search(smiles, sym):
    atoms, bonds = osm.molecule(smiles)
    want = osm.self_matches(atoms, bonds, sym)
    s = SmilesSet([smiles], perceive_aromaticity=True)
    got = s.self_matches(0, symmetrize_terminal_groups=sym)
    assert sorted(tuple(int(x) for x in r) for r in got) == want


def test_the_cap_on_the_number_of_matches_and_every_match_keeps_the_bonds():
    s = SmilesSet(["CC(C)(C)CC(C)(C)CC(C)(C)C", "c1ccc2cc3ccccc3cc2c1"], perceive_aromaticity=True)
    capped = s.self_matches(0, max_matches=1000)       # 3! 2! 3! for the methyls x 2 end for end: the whole group fits
    assert len(capped) == 144
    few = s.self_matches(0, max_matches=7)
    assert few.shape[0] == 7 and (few == capped[:7]).all()
    anth = s.self_matches(1)
    assert len(anth) == 4
    atoms, bonds = s.graph(1)
    have = {(int(a), int(b)): int(t) for a, b, t, _ in bonds} | {(int(b), int(a)): int(t) for a, b, t, _ in bonds}
    for r in anth:
        assert all(have.get((int(r[a]), int(r[b]))) == t for (a, b), t in have.items())


def test_a_complete_list_is_not_reported_as_truncated():
    """NVMK_TRUNCATED is the notice for a search abandoned on its step budget; complete lists and lists cut by max_matches are not."""
    s = SmilesSet(["c1ccccc1", "CC(C)(C)CC(C)(C)CC(C)(C)C"], perceive_aromaticity=True)
    m, truncated = s.self_matches(0, return_truncated=True)
    assert len(m) == 12 and truncated is False
    m, truncated = s.self_matches(1, max_matches=7, return_truncated=True)
    assert len(m) == 7 and truncated is False


def test_bad_arguments_are_refused():
    s = SmilesSet(["CC"], perceive_aromaticity=True)
    with pytest.raises(Exception):
        s.self_matches(5)
    with pytest.raises(Exception):
        s.self_matches(0, max_matches=0)


def test_matches_of_real_molecules_form_a_group():
    """The self matches of a molecule are its automorphism group: on the first molecules of the reference's chembl_10k.smi the
    returned set holds the identity, is closed under composition and under inversion, and every member keeps elements, charges,
    isotopes and bond types (checked against the graph the ingestion returns) — unless the 1000-match cap cut the set short."""
    path = ROOT / "tests" / "golden" / "chembl_10k.smi"
    lines = [ln.split()[0] for ln in path.read_text().splitlines()[:400] if ln.strip()]
    s = SmilesSet(lines, perceive_aromaticity=True)
    checked = symmetric = 0
    for i in range(len(lines)):
        if s.status[i] != 0 or s.n_atoms[i] > 60:
            continue
        m = s.self_matches(i, symmetrize_terminal_groups=False)
        n = int(s.n_atoms[i])
        assert m.shape[1] == n and (m[0] == np.arange(n)).all()
        if len(m) == 1000:
            continue                                    # capped: closure cannot be expected
        members = {tuple(int(x) for x in r) for r in m}
        assert len(members) == len(m)
        atoms, bonds = s.graph(i)
        have = {(int(a), int(b)): int(t) for a, b, t, _ in bonds} | {(int(b), int(a)): int(t) for a, b, t, _ in bonds}
        for r in m:
            assert (atoms[r, :3] == atoms[:, :3]).all()                                   # element, charge, isotope
            assert all(have.get((int(r[a]), int(r[b]))) == t for (a, b), t in have.items())
            inv = np.empty(n, dtype=np.int64)
            inv[r] = np.arange(n)
            assert tuple(int(x) for x in inv) in members
        if len(m) <= 48:
            for p in m:
                for q in m:
                    assert tuple(int(x) for x in p[q]) in members
        checked += 1
        symmetric += len(m) > 1
    assert checked > 200 and symmetric > 60, (checked, symmetric)
