"""RDKit-free ingestion of the fingerprint path: the library's SMILES parser
(nvmolkit_amd/csrc/smiles.cpp) against hand-computed graphs, against the independent restatement oracle/smiles.py on the
1000 ChEMBL SMILES the reference's tests read, and — through the Morgan oracle — against the element-count known answers
RDKit's test-suite holds for four SMILES (reference tests/test_morgan_fingerprint_ref.cpp:44-69)."""

from pathlib import Path

import numpy as np
import pytest

import oracle
from nvmolkit_amd.fingerprints import SMILES_STATUS, MorganFingerprintGenerator, SmilesSet
from oracle import smiles as osmi

CHEMBL = [line.split()[0] for line in (Path(__file__).parent / "golden" / "chembl_1k.smi").read_text().splitlines()
          if line.strip() and not line.startswith("#")]


def graph(smi):
    s = SmilesSet([smi])
    assert s.status[0] == 0, SMILES_STATUS[int(s.status[0])]
    return s.graph(0)


def hash_components(comps, ring):
    out = []
    for c, r in zip(comps.tolist(), ring.tolist()):
        v = [int(x) & 0xFFFFFFFF for x in c]
        out.append(oracle.morgan_hash_vector(v + [1] if r else v))
    return np.array(out, dtype=np.uint32)


def oracle_inputs(smiles_list, stride):
    """The five Morgan input arrays derived entirely on the oracle side."""
    n = len(smiles_list)
    atom_inv = np.zeros((n, stride), dtype=np.uint32)
    bond_inv = np.zeros((n, stride), dtype=np.uint32)
    bond_idx = np.full((n, stride, 8), -1, dtype=np.int16)
    bond_other = np.full((n, stride, 8), -1, dtype=np.int16)
    n_atoms = np.zeros(n, dtype=np.int16)
    for m, smi in enumerate(smiles_list):
        atoms, bonds = osmi.molecule(smi)
        comps, ring = osmi.invariant_components(atoms, bonds)
        n_atoms[m] = len(atoms)
        atom_inv[m, :len(atoms)] = hash_components(comps, ring)
        deg = np.zeros(len(atoms), dtype=int)
        for k, (a, b, t, _) in enumerate(bonds):
            bond_inv[m, k] = t
            for x, y in ((a, b), (b, a)):
                bond_idx[m, x, deg[x]] = k
                bond_other[m, x, deg[x]] = y
                deg[x] += 1
    return atom_inv, bond_inv, bond_idx, bond_other, n_atoms


# ---- hand-computed graphs ---------------------------------------------------------------------------
def test_ethanol_by_hand():
    atoms, bonds = graph("CCO")
    #                       Z  q iso H arom ring
    assert atoms.tolist() == [[6, 0, 0, 3, 0, 0], [6, 0, 0, 2, 0, 0], [8, 0, 0, 1, 0, 0]]
    assert bonds.tolist() == [[0, 1, 1, 0], [1, 2, 1, 0]]


def test_aromatic_rings_by_hand():
    atoms, bonds = graph("c1ccccc1")
    assert atoms.tolist() == [[6, 0, 0, 1, 1, 1]] * 6
    assert sorted(b[2] for b in bonds.tolist()) == [12] * 6 and all(b[3] == 1 for b in bonds.tolist())
    atoms, _ = graph("c1cc[nH]c1")                       # pyrrole: the hydrogen is written, aromatic n takes none by itself
    assert atoms[:, 3].tolist() == [1, 1, 1, 1, 1] and atoms[3, 0] == 7
    atoms, _ = graph("c1ccncc1")                          # pyridine: no hydrogen on n
    assert atoms[3].tolist() == [7, 0, 0, 0, 1, 1]
    atoms, bonds = graph("Cn1ccnc1")                      # N-methyl imidazole: substituted aromatic n, valence 4 > 3 -> no H
    assert atoms[1].tolist() == [7, 0, 0, 0, 1, 1] and bonds[0].tolist() == [0, 1, 1, 0]
    atoms, _ = graph("O=c1cc[nH]cc1")                     # pyridone: exocyclic C=O on an aromatic carbon
    assert atoms[1].tolist() == [6, 0, 0, 0, 1, 1]
    atoms, _ = graph("c1ccc2ccccc2c1")                    # naphthalene: fusion carbons (three aromatic bonds = 4.5) take no H
    assert atoms[:, 3].tolist() == [1, 1, 1, 0, 1, 1, 1, 1, 0, 1]
    atoms, _ = graph("c1ccsc1")                           # thiophene
    assert atoms[3].tolist() == [16, 0, 0, 0, 1, 1]


def test_biphenyl_link_is_single_with_or_without_the_dash():
    for smi in ("c1ccccc1-c1ccccc1", "c1ccccc1c1ccccc1"):
        _, bonds = graph(smi)
        link = [b for b in bonds.tolist() if b[3] == 0]
        assert len(link) == 1 and link[0][2] == 1
        assert sum(1 for b in bonds.tolist() if b[2] == 12) == 12


def test_hydrogens_charges_isotopes_and_fragments():
    atoms, bonds = graph("[H]C([H])([H])O")               # written hydrogens fold into the carbon
    assert atoms.tolist() == [[6, 0, 0, 3, 0, 0], [8, 0, 0, 1, 0, 0]] and len(bonds) == 1
    atoms, bonds = graph("[2H]C")                         # a labelled hydrogen stays an atom
    assert atoms.tolist() == [[1, 0, 2, 0, 0, 0], [6, 0, 0, 3, 0, 0]] and len(bonds) == 1
    atoms, bonds = graph("CC(=O)[O-].[Na+]")
    assert atoms.tolist() == [[6, 0, 0, 3, 0, 0], [6, 0, 0, 0, 0, 0], [8, 0, 0, 0, 0, 0], [8, -1, 0, 0, 0, 0], [11, 1, 0, 0, 0, 0]]
    assert bonds[:, 2].tolist() == [1, 2, 1]
    atoms, _ = graph("C[N+](=O)[O-]")
    assert atoms[1].tolist() == [7, 1, 0, 0, 0, 0]
    atoms, _ = graph("[NH4+]")
    assert atoms.tolist() == [[7, 1, 0, 4, 0, 0]]
    atoms, _ = graph("CS(=O)(=O)N")                       # sulfonamide: S valence 6
    assert atoms[1].tolist() == [16, 0, 0, 0, 0, 0] and atoms[4, 3] == 2
    atoms, _ = graph("OP(=O)(O)O")                        # phosphate: P valence 5
    assert atoms[1, 3] == 0
    atoms, _ = graph("C%10CC%10")                         # two-digit ring labels
    assert atoms[:, 5].tolist() == [1, 1, 1]
    atoms, _ = graph("C[C@@H](N)C(=O)O")                  # chirality marks are read and dropped
    assert atoms[1].tolist() == [6, 0, 0, 1, 0, 0]
    atoms, _ = graph("F/C=C/F")                           # directional bonds are single bonds
    assert atoms[:, 3].tolist() == [0, 1, 1, 0]


def test_hydrogens_rdkit_keeps_as_atoms():
    """RDKit's removeHs defaults: a hydrogen that defines double-bond stereo (its bond carries a direction, the
    neighbour a double bond) and a hydrogen on a dummy atom stay atoms; a hydrogen drawn on an aromatic atom written without
    brackets is that atom's hydrogen ('[H]n1cccc1' is pyrrole); product and oracle restatement agree on all of them."""
    atoms, bonds = graph("[H]/N=C(\\C)c1ccccc1")
    assert len(atoms) == 10 and atoms[0].tolist() == [1, 0, 0, 0, 0, 0] and atoms[1].tolist() == [7, 0, 0, 0, 0, 0]
    atoms, _ = graph("[H]N=C(C)c1ccccc1")                  # no direction: folded as usual
    assert len(atoms) == 9 and atoms[0].tolist() == [7, 0, 0, 1, 0, 0]
    atoms, _ = graph("[H]/N(C)C")                          # a direction without a double bond defines nothing
    assert len(atoms) == 3 and atoms[0, 3] == 1
    atoms, _ = graph("*[H]")
    assert atoms[:, 0].tolist() == [0, 1]
    s = SmilesSet(["[H]n1cccc1", "c1cc[nH]c1", "[H]c1ccccc1", "c1ccccc1"], perceive_aromaticity=False)
    assert s.status.tolist() == [0, 0, 0, 0]
    a0, a1, a2, a3 = (s.graph(i)[0] for i in range(4))
    assert sorted(map(tuple, a0.tolist())) == sorted(map(tuple, a1.tolist())) and np.array_equal(a2, a3)
    for smi in ("[H]/N=C(\\C)c1ccccc1", "[H]N=C(C)c1ccccc1", "[H]/N(C)C", "*[H]", "[H]n1cccc1", "[H]c1ccccc1", "[H]/C=C/[H]"):
        for got, want in zip(graph(smi), osmi.molecule(smi)):
            assert np.array_equal(got, want), smi


@pytest.mark.parametrize("smi,code", [("[C+99999999999999]", 1), ("[C+4294967297]", 1), ("[C+16]", 1), ("[C-15]", 2), ("[Fe+3]", 0),
                                      # isotopes: the table, labels whose truncated mass difference cannot depend on the mass
                                      # defect, and those where it could ([75Se] -3 / -4 ...), which are refused
                                      ("[13CH4]", 0), ("[2H]O[2H]", 0), ("[125I]C", 0), ("[75Se]", 0), ("[99Tc]", 0), ("[177Lu]", 0),
                                      ("[211At]", 6), ("[223Ra]", 6), ("[76Se]", 0), ("[18OH2]", 0), ("[11CH4]", 0), ("[57Co]", 0),
                                      ("[64Cu]", 0), ("[68Ga]", 0), ("[89Zr]", 0), ("[111In]", 0), ("[153Gd]", 0), ("[201Tl]", 0),
                                      # within 0.005 u of an integer step of mass - weight: depends on the last digit of the weight table
                                      ("[60Co]", 6), ("[90Y]", 6), ("[137Cs]", 6)])
def test_charge_digits_and_isotope_labels(smi, code):
    assert int(SmilesSet([smi]).status[0]) == code


def test_isotope_invariants_where_the_mass_number_alone_would_be_wrong():
    """int(mass - average weight) with the exact masses (RDKit gives -4 for [75Se], 0 for [99Tc]; the mass number
    alone gives -3 and 1): 74.9225 - 78.971, 98.9063 - 98, 75.9192 - 78.971, 63.9298 - 63.546, 88.9089 - 91.224."""
    import zlib  # noqa: F401
    want = {"[75Se]": -4, "[99Tc]": 0, "[76Se]": -3, "[64Cu]": 0, "[68Ga]": -1, "[89Zr]": -2, "[111In]": -3, "[153Gd]": -4,
            "[201Tl]": -3, "[57Co]": -1, "[177Lu]": 1, "[13CH4]": 0, "[14CH4]": 1, "[2H]C": 1}
    for smi, d in want.items():
        s = SmilesSet([smi])
        assert s.status[0] == 0
        z = int(s.graph(0)[0][0, 0])
        ai, *_ = s.morgan_inputs([0], 32)
        atoms = s.graph(0)[0]
        comps = [int(atoms[0, 0]), int(atoms[0, 3]), int(atoms[0, 3]) + (1 if smi == "[2H]C" else 0), int(atoms[0, 1]), d]
        if smi == "[2H]C":  # the hydrogen atom itself: degree 1, no hydrogens
            comps = [1, 1, 0, 0, d]
        seed = 0
        for c in comps:
            seed = (seed ^ ((c & 0xffffffff) + 0x9e3779b9 + ((seed << 6) & 0xffffffff) + (seed >> 2))) & 0xffffffff
        assert int(ai[0, 0]) & 0xffffffff == seed, (smi, z)


def test_labelled_dummy_atoms_take_the_label_as_their_mass():
    """a dummy atom has no nuclide, its isotope label is the mass itself (RDKit: the label as a double, weight 0) —
    the band of mass excesses applies to real elements only.  Product == oracle/smiles.py == the hand-computed invariant."""
    for smi, label in [("[25*]C", 25), ("[120*]C", 120), ("[16*]C", 16), ("[30*]C", 30), ("[1*]C", 1), ("[213*]C", 213)]:
        s = SmilesSet([smi])
        assert s.status[0] == 0, smi
        ai, *_ = s.morgan_inputs([0], 32)
        comps, _ = osmi.invariant_components(*osmi.molecule(smi))
        assert comps[0].tolist() == [0, 1, 0, 0, label], smi
        seed = 0
        for c in comps[0].tolist():
            seed = (seed ^ ((c & 0xffffffff) + 0x9e3779b9 + ((seed << 6) & 0xffffffff) + (seed >> 2))) & 0xffffffff
        assert int(ai[0, 0]) & 0xffffffff == seed, smi


def test_ring_labels_beyond_99_and_elements_by_atomic_number():
    """Two spellings RDKit's SMILES parser reads beyond OpenSMILES: %(n) ring-closure labels (up to five digits) and [#n] atoms."""
    for written, plain in [("C%(100)CC%(100)", "C1CC1"), ("C%(7)CC7", "C1CC1"), ("C%(12345)CC%(12345)O", "C1CC1O"),
                           ("c%(100)ccccc%(100)", "c1ccccc1"), ("C%(100)CC%(100)C%(100)CC%(100)", "C1CC1C1CC1"),
                           ("[#6]", "[C]"), ("[#6H4]", "[CH4]"), ("[#7+]([#8-])(=O)c1ccccc1", "[N+]([O-])(=O)c1ccccc1"),
                           ("[#0]C", "*C"), ("[13#6H4]", "[13CH4]"), ("[#6H2]1[#6H2][#8]1", "[CH2]1[CH2][O]1")]:
        for got, want in zip(graph(written), graph(plain)):
            assert np.array_equal(got, want), written
        for got, want in zip(graph(written), osmi.molecule(written)):
            assert np.array_equal(got, want), written
    refused = SmilesSet(["C%(123456)CC%(123456)", "C%(100)CC", "C%()CC", "C%(1x)CC%(1x)", "[#119]", "[#]", "[#6", "C%(100"])
    assert refused.status.tolist() == [1] * 8


def test_ring_membership_is_cycle_membership():
    atoms, bonds = graph("C1CC1CC1CCC1")                  # two rings joined by a CH2: the linker is on no cycle
    assert atoms[:, 5].tolist() == [1, 1, 1, 0, 1, 1, 1, 1]
    assert bonds[:, 3].tolist() == [1, 1, 1, 0, 0, 1, 1, 1, 1]
    atoms, _ = graph("C1CC2CCC1C2")                       # bicyclic: every atom is on a cycle
    assert atoms[:, 5].tolist() == [1] * 7


@pytest.mark.parametrize("smi,code", [("C(", 1), ("C1CC", 1), ("C)", 1), ("[Xx]", 1), ("C=", 1), ("C1C1", 1), ("", 0),
                                      ("CN(=O)=O", 0), ("C(C)(C)(C)(C)C", 2), ("OCl(=O)(=O)=O", 0), ("CN(C)(C)(C)C", 2), ("FCl(=O)=O", 2),
                                      ("C1=CC=CC=C1", 3), ("C1=CNC=C1", 3), ("C1=COC=C1", 3), ("C1=CC=C2C=CC=CC2=C1", 3),
                                      ("C1=CCCCC1", 0), ("O=C1C=CC(=O)C=C1", 0), ("C1=CC=CC1", 0), ("C1=CC=CCC=C1", 0),
                                      # bracket atoms are valence-checked too (highest valence shifted by the charge)
                                      ("[CH5]", 2), ("C[N](C)(C)C", 2), ("C[N+](C)(C)C", 0), ("[O](C)(C)C", 2), ("C[O+](C)C", 0),
                                      ("[B-](F)(F)(F)F", 0), ("[B](F)(F)(F)F", 2), ("[C-]#[O+]", 0), ("[CH3]", 0), ("[CH2-]C", 0),
                                      ("[C-](C)(C)(C)C", 2), ("[C+](C)(C)(C)C", 2), ("[NH4+]", 0), ("[NH4]", 2), ("[OH3+]", 0), ("[O-]C", 0), ("[O-](C)C", 2),
                                      # the two failures RDKit's "Getting Started" shows: 'CO(C)C' ("Explicit valence for atom # 1 O, 3,
                                      # is greater than permitted") and 'c1cc1' ("Can't kekulize mol")
                                      ("CO(C)C", 2), ("c1cc1", 5),
                                      # no Kekule structure (RDKit: "Can't kekulize mol"), aromatic marks outside rings
                                      ("c1cccc1", 5), ("c1ccnc1", 5), ("cc", 5), ("C:C", 5), ("c1cc[nH]c1", 0), ("c1ccn(C)c1", 0),
                                      # written aromatic where RDKit perceives none (cyclooctatetraene, 4-pyranone ring carbon chain)
                                      ("c1ccccccc1", 3)])
def test_refusals_and_their_neighbours(smi, code):
    assert int(SmilesSet([smi], perceive_aromaticity=False).status[0]) == code      # the strict mode


@pytest.mark.parametrize("written,clean", [
    ("CN(=O)=O", "C[N+]([O-])=O"), ("C1=CC=CN(=O)=C1", "C1=CC=C[N+]([O-])=C1"), ("CN=N#N", "CN=[N+]=[N-]"),
    ("C=P(=O)O", "C=[P+]([O-])O"), ("O=Cl(=O)O", "[O-][Cl+2]([O-])O"), ("O=N(=O)c1ccccc1", "[O-][N+](=O)c1ccccc1"),
    ("OCl(=O)(=O)=O", "O[Cl+3]([O-])([O-])[O-]"), ("c1cccn(=O)c1", "c1ccc[n+]([O-])c1")])
def test_hypervalent_spellings_are_cleaned_up_like_rdkit(written, clean):
    """The examples of the RDKit Book's "Sanitization" section (MolOps::cleanUp): both spellings are one molecule."""
    a, b = SmilesSet([written]), SmilesSet([clean])
    assert a.status[0] == 0 and b.status[0] == 0
    (aa, ab), (ba, bb) = a.graph(0), b.graph(0)
    assert np.array_equal(aa, ba) and np.array_equal(ab, bb), (aa.tolist(), ba.tolist())
    strict = SmilesSet([written], perceive_aromaticity=False)        # the oracle reads aromaticity as written
    if strict.status[0] == 0:
        for x, y in zip(osmi.molecule(written), strict.graph(0)):
            assert np.array_equal(x, y)


# ---- against the independent restatement --------------------------------------------------------------
def test_graphs_equal_the_oracle_on_chembl_1k():
    s = SmilesSet(CHEMBL)
    assert len(CHEMBL) == 999 and np.all(s.status == 0)
    for i, smi in enumerate(CHEMBL):
        atoms, bonds = s.graph(i)
        want_atoms, want_bonds = osmi.molecule(smi)
        assert np.array_equal(atoms, want_atoms), smi
        assert np.array_equal(bonds, want_bonds), smi


def test_morgan_inputs_equal_the_oracle_on_chembl_1k():
    s = SmilesSet(CHEMBL)
    size = np.maximum(s.n_atoms, s.n_bonds)
    idx = np.flatnonzero(size < 64)[:300]
    got = s.morgan_inputs(idx, 64)
    want = oracle_inputs([CHEMBL[i] for i in idx], 64)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_parallel_and_serial_parsing_agree():
    a, b = SmilesSet(CHEMBL, 1), SmilesSet(CHEMBL, 8)
    assert np.array_equal(a.n_atoms, b.n_atoms) and np.array_equal(a.n_bonds, b.n_bonds) and np.array_equal(a.status, b.status)
    ids = np.flatnonzero(np.maximum(a.n_atoms, a.n_bonds) < 128)
    for x, y in zip(a.morgan_inputs(ids, 128, 1), b.morgan_inputs(ids, 128, 8)):
        assert np.array_equal(x, y)


def test_morgan_inputs_refuse_what_was_not_ingested():
    s = SmilesSet(["CCO", "C1=CC=CC=C1", "CCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCC"], perceive_aromaticity=False)
    with pytest.raises(ValueError, match="not ingested"):
        s.morgan_inputs([0, 1], 32)
    with pytest.raises(ValueError, match="does not fit"):
        s.morgan_inputs([2], 32)
    assert s.morgan_inputs([0], 32)[4].tolist() == [3]


# ---- RDKit's own known answers, through SMILES ---------------------------------------------------------
@pytest.mark.parametrize("smi,want", [("CCCCC", [2, 5, 7, 7]), ("O=C(O)CC1CC1", [6, 12, 16, 17]), ("OC(=O)CC1CC1", [6, 12, 16, 17])])
def test_environment_counts_known_answers_from_smiles(smi, want):
    # tests/test_morgan_fingerprint_ref.cpp:53-58: number of distinct sparse Morgan ids per radius 0..3
    ai, bi, bx, bo, na = SmilesSet([smi]).morgan_inputs([0], 32)
    for radius, n_expected in enumerate(want):
        codes, _ = oracle.morgan_environments(ai[0], bi[0], bx[0], bo[0], int(na[0]), radius)
        assert len(set(codes.tolist())) == n_expected, f"radius {radius}"


def test_symmetry_known_answer_from_smiles():
    # :60-69: OCCCCO, radius 2 -> 7 distinct ids, each seen 2 or 4 times
    ai, bi, bx, bo, na = SmilesSet(["OCCCCO"]).morgan_inputs([0], 32)
    codes, _ = oracle.morgan_environments(ai[0], bi[0], bx[0], bo[0], int(na[0]), 2)
    values, counts = np.unique(codes, return_counts=True)
    assert len(values) == 7 and set(counts.tolist()) <= {2, 4}


def test_atom_order_invariance_from_smiles():
    fps = []
    for smi in ("O=C(O)CC1CC1", "OC(=O)CC1CC1", "C1CC1CC(O)=O", "C(C1CC1)C(=O)O"):
        ai, bi, bx, bo, na = SmilesSet([smi]).morgan_inputs([0], 32)
        fps.append(oracle.morgan_fingerprints(ai, bi, bx, bo, na, 32, 2, 2048))
    assert all(np.array_equal(fps[0], f) for f in fps[1:])


# ---- text buffers, chunked storage, staging blocks (host only) ------------------------------------------
def _same_sets(a, b):
    assert len(a) == len(b)
    assert np.array_equal(a.n_atoms, b.n_atoms) and np.array_equal(a.n_bonds, b.n_bonds) and np.array_equal(a.status, b.status)
    for i in range(0, len(a), max(1, len(a) // 97)):
        for x, y in zip(a.graph(i), b.graph(i)):
            assert np.array_equal(x, y)


def test_text_buffers_are_parsed_line_by_line():
    path = Path(__file__).parent / "golden" / "chembl_1k.smi"
    lines = path.read_text().split("\n")
    lines = lines[:-1] if lines[-1] == "" else lines                                           # a final newline ends the last line
    listed = SmilesSet([line.split()[0] if line.split() else "" for line in lines])
    _same_sets(SmilesSet.from_file(path), listed)
    _same_sets(SmilesSet.from_text(path.read_text().replace("\n", "\r\n")), listed)          # DOS line ends
    _same_sets(SmilesSet.from_text(path.read_bytes().rstrip(b"\n")), listed)                   # no terminator after the last line
    _same_sets(SmilesSet.from_text(path.read_bytes().rstrip(b"\n") + b"\n"), listed)
    # the pointer-array entry (strings with a line break inside) and the text entry see the same molecules
    _same_sets(SmilesSet(CHEMBL + ["CC\nO"]), SmilesSet(CHEMBL + ["CC"]))
    # every line counts: names are ignored, an empty line is an empty molecule, a header is a syntax error
    s = SmilesSet.from_text("smiles name\nCCO ethanol\n\nc1ccccc1\tbenzene\n  CC")
    assert s.n_atoms.tolist() == [0, 3, 0, 6, 2] and s.status.tolist() == [1, 0, 0, 0, 0]
    assert len(SmilesSet.from_text("")) == 0 and len(SmilesSet.from_text("\n")) == 1 and len(SmilesSet([""])) == 1
    assert SmilesSet(["C", "", "CC", ""]).n_atoms.tolist() == [1, 0, 2, 0]


def test_chunk_boundaries_keep_molecule_order():
    """Molecules are stored in chunks of 512 filled by different threads: molecule i must stay string i."""
    smiles = ["C" * (1 + i % 37) + ("O" if i % 3 else "N") for i in range(512 * 5 + 17)]
    for threads in (1, 3, 8):
        s = SmilesSet(smiles, threads)
        assert s.n_atoms.tolist() == [len(x) for x in smiles]
        for i in (0, 511, 512, 513, 1023, 1024, 2559, 2560, len(smiles) - 1):
            atoms, _ = s.graph(i)
            assert atoms[-1, 0] == (8 if i % 3 else 7) and len(atoms) == len(smiles[i])


def test_staging_block_holds_the_same_arrays():
    from nvmolkit_amd import fingerprints as fpmod
    s = SmilesSet(CHEMBL)
    size = np.maximum(s.n_atoms, s.n_bonds)
    idx = np.flatnonzero((size >= 32) & (size < 64) & (s.status == 0))
    sizes, specs = fpmod._smiles_block_sizes(len(idx), 64)
    offsets, total = fpmod._block_layout(sizes)
    assert all(o % 256 == 0 for o in offsets) and total >= sum(sizes)
    block = np.full(total + 100, 0xAB, dtype=np.uint8)
    fpmod._fill_smiles_block(block, offsets, s, idx, 64, 4)
    want = list(s.morgan_inputs(idx, 64)) + [idx.astype(np.int32)]
    for o, size_b, (shape, dtype), w in zip(offsets, sizes, specs, want):
        assert np.array_equal(block[o:o + size_b].view(dtype).reshape(shape), w)
    assert (block[total:] == 0xAB).all()
    with pytest.raises(ValueError, match="out arrays"):
        s.morgan_inputs(idx, 64, out=tuple(np.zeros((1, 1), dtype=np.uint32) for _ in range(5)))


def _stub_the_device(monkeypatch):
    """Replaces everything CUDA of the fingerprint module by host equivalents: device tensors become host tensors, the pinned
    pool plain memory, streams nothing, and nvmk_morgan_from_invariants is answered by the CPU oracle on the staged arrays.
    Returns the list the stubbed kernel appends (max_atoms, n_mols) to."""
    import contextlib
    import ctypes

    import torch

    from nvmolkit_amd import _native
    from nvmolkit_amd import fingerprints as fpmod
    real = _native.lib()
    calls = []

    class Stub:
        def __getattr__(self, name):
            return getattr(real, name)

        @staticmethod
        def nvmk_morgan_from_invariants(p_ai, p_bi, p_bx, p_bo, p_na, p_rows, n, max_atoms, radius, fp_bits, p_out, stream):
            def arr(ptr, count, ctype, dtype):
                return np.frombuffer((ctype * count).from_address(ptr), dtype=dtype).copy()
            ai = arr(p_ai, n * max_atoms, ctypes.c_uint32, np.uint32).reshape(n, max_atoms)
            bi = arr(p_bi, n * max_atoms, ctypes.c_uint32, np.uint32).reshape(n, max_atoms)
            bx = arr(p_bx, n * max_atoms * 8, ctypes.c_int16, np.int16).reshape(n, max_atoms, 8)
            bo = arr(p_bo, n * max_atoms * 8, ctypes.c_int16, np.int16).reshape(n, max_atoms, 8)
            na = arr(p_na, n, ctypes.c_int16, np.int16)
            rows = arr(p_rows, n, ctypes.c_int32, np.int32) if p_rows else np.arange(n, dtype=np.int32)
            fp = oracle.morgan_fingerprints(ai, bi, bx, bo, na, max_atoms, radius, fp_bits)
            words = fp_bits // 32
            for r, row in zip(rows.tolist(), fp):
                dst = (ctypes.c_uint32 * words).from_address(p_out + 4 * words * r)
                np.frombuffer(dst, dtype=np.uint32)[:] = row
            calls.append((max_atoms, n))
            return 0

    real_zeros = torch.zeros
    monkeypatch.setattr(_native, "lib", lambda: Stub())
    monkeypatch.setattr(_native, "on_stream", lambda stream, dev: contextlib.nullcontext())
    monkeypatch.setattr(_native, "stream_ptr", lambda stream: 0)
    monkeypatch.setattr(fpmod, "_pinned_block", lambda n: torch.empty(n, dtype=torch.uint8))
    monkeypatch.setattr(fpmod, "_release_pinned_block", lambda *a: None)
    monkeypatch.setattr(torch.cuda, "device", lambda dev: contextlib.nullcontext())
    monkeypatch.setattr(torch, "zeros", lambda *a, **kw: real_zeros(*a, **{k: v for k, v in kw.items() if k != "device"}))
    return calls


def test_smiles_to_fingerprints_host_path_with_the_kernel_call_stubbed(monkeypatch):
    """Everything of GetFingerprintsFromSmiles' bucket path that is not CUDA — bucketing, the staging block, the pointers and
    the output rows handed to nvmk_morgan_from_invariants — with that one call answered by the oracle on the staged arrays
    (the real kernel is compared with the same expectation in the GPU test below)."""
    import torch

    calls = _stub_the_device(monkeypatch)
    gen = MorganFingerprintGenerator(radius=2, fpSize=2048)
    mols = SmilesSet(CHEMBL)
    size = np.maximum(mols.n_atoms, mols.n_bonds)
    out = torch.zeros((len(mols), 64), dtype=torch.int32)
    lo = 0
    for b in (32, 64, 128, 256, 512, 1024):
        idx = np.flatnonzero((size >= lo) & (size < b) & (mols.status == 0))
        lo = b
        if len(idx):
            gen._launch_smiles(mols, idx, b, out, 0, None)
            # the array-input seam goes through the same submit step
            check = torch.zeros((len(mols), 64), dtype=torch.int32)
            gen._launch(mols.morgan_inputs(idx, b), b, check, idx, None)
            assert torch.equal(check[idx], out[idx])
    got = out.numpy().view(np.uint32)
    assert len(calls) >= 6 and (got != 0).any(axis=1).all()
    sizes = np.array([max(len(a), len(b)) for a, b in (osmi.molecule(smi) for smi in CHEMBL)])
    lo = 0
    for stride in (32, 64, 128, 256, 512, 1024):
        idx = np.flatnonzero((sizes >= lo) & (sizes < stride))
        lo = stride
        if len(idx):
            assert np.array_equal(got[idx], oracle.morgan_fingerprints(*oracle_inputs([CHEMBL[i] for i in idx], stride), stride, 2, 2048))


def test_the_gpu_tests_of_this_file_on_the_stubbed_device(monkeypatch):
    """The bodies of the three GPU tests below, run against the stub: what they exercise beyond the kernel (dispatch of string
    lists, refusals, zero rows, SD files, repeated generators) is host code and is checked here on every CPU run as well."""
    calls = _stub_the_device(monkeypatch)
    test_fingerprints_from_smiles_equal_the_oracle_pipeline()
    test_refused_smiles_raise_or_stay_zero()
    from tests import test_benchmark_molecules_gpu as late
    late.test_repeated_single_molecule_calls_never_come_back_empty()
    late.test_kekule_and_aromatic_spellings_give_one_fingerprint()
    late.test_kernel_bits_of_the_documented_examples()
    assert len(calls) > 256
    # molecules read from an SD file take the same route
    sdf = SmilesSet.from_sdf_file(Path(__file__).parent / "golden" / "larger_molecules.sdf")
    fp = MorganFingerprintGenerator(radius=2, fpSize=1024).GetFingerprintsFromSmiles(sdf).torch().numpy().view(np.uint32)
    assert np.array_equal(fp, oracle.morgan_fingerprints(*sdf.morgan_inputs([0, 1, 2], 64), 64, 2, 1024))


# ---- on the GPU ----------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_fingerprints_from_smiles_equal_the_oracle_pipeline():
    gen = MorganFingerprintGenerator(radius=2, fpSize=2048)
    got = gen.GetFingerprints(CHEMBL).torch().cpu().numpy().view(np.uint32)   # strings dispatch to the SMILES path
    sizes = []
    for smi in CHEMBL:
        a, b = osmi.molecule(smi)
        sizes.append(max(len(a), len(b)))
    sizes = np.array(sizes)
    lo = 0
    for stride in (32, 64, 128, 256, 512, 1024):
        idx = np.flatnonzero((sizes >= lo) & (sizes < stride))
        lo = stride
        if len(idx):
            want = oracle.morgan_fingerprints(*oracle_inputs([CHEMBL[i] for i in idx], stride), stride, 2, 2048)
            assert np.array_equal(got[idx], want)
    assert (got != 0).any(axis=1).all()


@pytest.mark.gpu
def test_refused_smiles_raise_or_stay_zero():
    gen = MorganFingerprintGenerator(radius=2, fpSize=1024)
    with pytest.raises(ValueError, match="1 of 3 SMILES were not ingested"):
        gen.GetFingerprintsFromSmiles(["CCO", "C1=CC=CC=C1", "c1ccccc1"], perceive_aromaticity=False)
    res = gen.GetFingerprintsFromSmiles(["CCO", "C1=CC=CC=C1", "c1ccccc1"], on_error="zero", perceive_aromaticity=False)
    fp = res.torch().cpu().numpy()
    assert res.smiles_status.tolist() == [0, 3, 0] and not fp[1].any() and fp[0].any() and fp[2].any()


BINAP_LIKE = "CC1(C)C2=C(C=CC(=C2)P(C3=CC=CC=C3)C4=CC=CC=C4)OC5=C1C=CC(=C5)P(C6=CC=CC=C6)C7=CC=CC=C7"
BINAP_LIKE_AROMATIC = "CC1(C)c2c(ccc(c2)P(c3ccccc3)c4ccccc4)Oc5c1ccc(c5)P(c6ccccc6)c7ccccc7"


def test_kekule_input_of_the_reference_regression_molecule():
    """The molecule of the reference's regression test (nvmolkit/tests/test_fingerprints.py:137-148) is written in Kekule
    form: refused in the strict mode, perceived otherwise, and then the same graph as its aromatic form."""
    assert SmilesSet([BINAP_LIKE], perceive_aromaticity=False).status[0] == 3
    a, b = SmilesSet([BINAP_LIKE], perceive_aromaticity=True), SmilesSet([BINAP_LIKE_AROMATIC])
    assert a.status[0] == 0 and b.status[0] == 0
    for x, y in zip(a.graph(0), b.graph(0)):
        assert np.array_equal(x, y)
    assert int((a.graph(0)[1][:, 2] == 12).sum()) == 36


def test_mutated_smiles_never_crash_and_agree_with_the_oracle():
    """Random edits of real SMILES (mostly invalid afterwards): the parser refuses or ingests exactly what the oracle does,
    with the same graph, and survives everything."""
    import random

    rng = random.Random(7)
    alphabet = "CNOPSFIclBrnos()[]=#-+123456789%@H/\\.:*$ "
    muts = []
    for _ in range(4000):
        s = list(rng.choice(CHEMBL)[:100])
        for _ in range(rng.randint(1, 4)):
            op, pos = rng.random(), rng.randrange(len(s) + 1)
            if op < 0.4 and s:
                s[min(pos, len(s) - 1)] = rng.choice(alphabet)
            elif op < 0.7:
                s.insert(pos, rng.choice(alphabet))
            elif s:
                del s[min(pos, len(s) - 1)]
        muts.append("".join(s))
    muts += ["", "[", "]", "[]", "[C", "C]", "%", "%1", "C%1", "((", "C((C))", "[12345C]", "[C" + "+" * 20 + "]", "[CH999]", "C" * 5000,
             "C1CC1C1CC1" * 50, "[*]", "*", "[C@O]", "[C@TH]", "[C@TH1](F)(Cl)(Br)I", " CCO", "CCO name", "C=#C", "C..C", ".C", "C."]
    got = SmilesSet(muts, perceive_aromaticity=False)        # strict mode: the graph is the written one, as the oracle reads it
    n_ok = 0
    for i, m in enumerate(muts):
        try:
            atoms, bonds = osmi.molecule(m)
            oracle_ok = True
        except osmi.SmilesError:
            oracle_ok = False
        st = int(got.status[i])
        assert oracle_ok == (st not in (1, 2)), (m, st)        # syntax and valence refusals coincide
        if oracle_ok and st == 0:
            ga, gb = got.graph(i)
            assert np.array_equal(ga, atoms) and np.array_equal(gb, bonds), m
            n_ok += 1
    assert n_ok > 100


def test_cfg1_fingerprints_equal_the_committed_digest():
    """The CPU half of BASELINE configs[0] (ingestion + oracle Morgan on the 10 000 benchmark SMILES) still gives the committed
    fingerprints (tests/golden/cfg1_chembl_10k_digest.json); the GPU half is held to the same file in test_config_size_gpu.py."""
    import hashlib
    import importlib.util
    import json

    golden_dir = Path(__file__).parent / "golden"
    spec = importlib.util.spec_from_file_location("make_cfg1_digest", golden_dir / "make_cfg1_digest.py")
    maker = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(maker)
    fp = maker.fingerprints(golden_dir / "chembl_10k.smi")
    golden = json.loads((golden_dir / "cfg1_chembl_10k_digest.json").read_text())
    assert hashlib.sha256(np.ascontiguousarray(fp).tobytes()).hexdigest() == golden["fingerprints_sha256"]
    assert int(np.unpackbits(fp.view(np.uint8), axis=1).sum()) == golden["bits_set_total"] == 523296
    assert sum(golden["similarity_histogram_floor_100x"]) == 10_000 ** 2 and golden["similarity_histogram_floor_100x"][100] >= 10_000
    assert maker.butina_digest(fp, 0.3) == golden["butina"]["0.3"] and golden["butina"]["0.3"]["clusters"] == 6592
