"""SD-file ingestion of the fingerprint path (SmilesSet.from_sdf_text / nvmk_sdf_parse_text; SURVEY.md 8(f) item 4 names
SMILES and SDF) on the molfiles the reference's own tests read: samples of tests/test_data/MMFF94_dative.sdf and
MMFF94_hypervalent.sdf (every fourth record) and its two files of larger molecules.  Those files draw EVERY hydrogen, so
the hydrogen counts the library derives from the valence model (charges, higher valence states) can be held against what
is drawn, atom by atom."""

from pathlib import Path

import numpy as np
import pytest

import oracle
from nvmolkit_amd.fingerprints import SmilesSet
from oracle import aromaticity as oarom
from oracle import molfile as omol

GOLDEN = Path(__file__).parent / "golden"
FILES = ["MMFF94_dative_every4th.sdf", "MMFF94_hypervalent_every4th.sdf", "larger_molecules.sdf"]


@pytest.mark.parametrize("name", FILES)
def test_graphs_and_hydrogen_counts_equal_what_the_file_draws(name):
    text = (GOLDEN / name).read_text()
    records = omol.records(text)
    got = SmilesSet.from_sdf_text(text, perceive_aromaticity=False)          # strict mode keeps the bond types as drawn
    assert len(got) == len(records) and set(got.status.tolist()) <= {0, 3}    # 3: a Kekule-form aromatic ring, as drawn
    charged = 0
    for i, rec in enumerate(records):
        _, z, q, iso, bonds = omol.read(rec)
        atoms, heavy_bonds, _ = omol.heavy_atom_graph(z, q, iso, bonds)
        ga, gb = got.graph(i)
        assert np.array_equal(ga[:, :4], atoms), rec[0]                       # Z, charge, isotope, hydrogens == drawn hydrogens
        assert np.array_equal(gb[:, :3], heavy_bonds), rec[0]
        charged += int((atoms[:, 1] != 0).any())
    assert charged > 20 or name == "larger_molecules.sdf"


@pytest.mark.parametrize("name", FILES)
def test_sdf_and_smiles_paths_give_the_same_molecules(name):
    """The heavy-atom graph of each record, written as a SMILES by the oracle and read by the SMILES entry, is the molecule
    the SD entry built — atoms, perceived aromaticity, bonds — and gives the same fingerprint (CPU oracle on both inputs)."""
    text = (GOLDEN / name).read_text()
    records = omol.records(text)
    sdf = SmilesSet.from_sdf_text(text)
    assert np.all(sdf.status == 0)
    smiles, orders = [], []
    for rec in records:
        _, z, q, iso, bonds = omol.read(rec)
        atoms, hb, _ = omol.heavy_atom_graph(z, q, iso, bonds)
        table = np.concatenate([atoms, np.zeros((len(atoms), 2), dtype=atoms.dtype)], 1)
        btable = np.concatenate([hb, np.zeros((len(hb), 1), dtype=np.int64)], 1)
        smi, order = oarom.write_bracket_smiles(table, hb[:, 2], btable)
        smiles.append(smi)
        orders.append(np.asarray(order))
    smi_set = SmilesSet(smiles)
    assert np.all(smi_set.status == 0)
    n_aromatic = 0
    for i, order in enumerate(orders):
        (sa, sb), (ma, mb) = sdf.graph(i), smi_set.graph(i)
        assert np.array_equal(sa[order], ma), records[i][0]                   # the string's j-th atom is atom order[j] of the record
        inv = np.empty(len(order), dtype=np.int64)
        inv[order] = np.arange(len(order))
        want = {(min(inv[a], inv[b]), max(inv[a], inv[b])): (t, r) for a, b, t, r in sb}
        have = {(min(a, b), max(a, b)): (t, r) for a, b, t, r in mb}
        assert want == have, records[i][0]
        n_aromatic += int((sb[:, 2] == 12).any())
    assert n_aromatic > 40 or name == "larger_molecules.sdf"
    size = np.maximum(sdf.n_atoms, sdf.n_bonds)
    ids = np.flatnonzero(size < 64)
    a = oracle.morgan_fingerprints(*sdf.morgan_inputs(ids, 64), 64, 2, 2048)
    b = oracle.morgan_fingerprints(*smi_set.morgan_inputs(ids, 64), 64, 2, 2048)
    assert np.array_equal(a, b) and a.any(axis=1).all()


def test_dative_and_hypervalent_files_hold_the_same_molecules():
    """The two files differ only in how S and P oxides (and S=N) are charged (RDKit's clean-up, applied when they were written, had
    already turned every nitro group into the charge-separated form in both): same atoms, hydrogens and connectivity."""
    d = SmilesSet.from_sdf_file(GOLDEN / FILES[0], perceive_aromaticity=False)
    h = SmilesSet.from_sdf_file(GOLDEN / FILES[1], perceive_aromaticity=False)
    same = 0
    for i in range(len(d)):
        (da, db), (ha, hb) = d.graph(i), h.graph(i)
        assert np.array_equal(da[:, [0, 2, 3]], ha[:, [0, 2, 3]]) and np.array_equal(db[:, :2], hb[:, :2])
        differs = np.flatnonzero(da[:, 1] != ha[:, 1])
        assert set(da[differs, 0].tolist()) <= {7, 8, 15, 16}                   # S=O, P=O, S=N written as dative bonds
        same += int(len(differs) == 0 and np.array_equal(db, hb))
    assert same > 100


def molblock(atoms, bonds, props=(), version="V2000", end=True):
    """atoms: (symbol, charge code) or (symbol, charge code, mass difference); bonds: (a, b, type), 1-based"""
    lines = ["made by hand", "  test", "", f"{len(atoms):3d}{len(bonds):3d}  0  0  0  0  0  0  0  0999 {version}"]
    for at in atoms:
        sym, ccc = at[0], at[1]
        dd = at[2] if len(at) > 2 else 0
        lines.append(f"{0.0:10.4f}{0.0:10.4f}{0.0:10.4f} {sym:<3s}{dd:2d}{ccc:3d}  0  0  0  0  0  0  0  0  0  0")
    lines += [f"{a:3d}{b:3d}{t:3d}  0" for a, b, t in bonds]
    lines += list(props)
    if end:
        lines.append("M  END")
    return "\n".join(lines) + "\n"


def one(block, **kw):
    s = SmilesSet.from_sdf_text(block + "$$$$\n", **kw)
    assert len(s) == 1
    return s


def test_hand_made_records():
    benzene = molblock([("C", 0)] * 6, [(i + 1, (i + 1) % 6 + 1, 4) for i in range(6)])
    s = one(benzene)
    assert s.status[0] == 0 and s.graph(0)[0][:, [0, 3, 4, 5]].tolist() == [[6, 1, 1, 1]] * 6 and set(s.graph(0)[1][:, 2]) == {12}
    kekule = molblock([("C", 0)] * 6, [(i + 1, (i + 1) % 6 + 1, 1 + i % 2) for i in range(6)])
    assert one(kekule, perceive_aromaticity=False).status[0] == 3 and np.array_equal(one(kekule).graph(0)[0], s.graph(0)[0])
    # hydrogens drawn on an atom in a higher valence state stay (H3P=O); elsewhere the valence model decides (CH2 drawn -> CH4)
    assert one(molblock([("P", 0), ("O", 0), ("H", 0), ("H", 0), ("H", 0)], [(1, 2, 2), (1, 3, 1), (1, 4, 1), (1, 5, 1)])).graph(0)[0][:, 3].tolist() == [3, 0]
    assert one(molblock([("C", 0), ("H", 0), ("H", 0)], [(1, 2, 1), (1, 3, 1)])).graph(0)[0][:, 3].tolist() == [4]
    # charges: the atom block's column, superseded by M  CHG lines; isotopes: mass difference, M  ISO, D and T
    assert one(molblock([("N", 3), ("C", 0)], [(1, 2, 1)])).graph(0)[0][:, [0, 1, 3]].tolist() == [[7, 1, 3], [6, 0, 3]]
    assert one(molblock([("N", 3), ("O", 0)], [(1, 2, 1)], ["M  CHG  1   2  -1"])).graph(0)[0][:, [0, 1, 3]].tolist() == [[7, 0, 2], [8, -1, 0]]
    assert one(molblock([("C", 0, 1), ("O", 0)], [(1, 2, 1)], ["M  ISO  1   2  18"])).graph(0)[0][:, 2].tolist() == [13, 18]
    assert one(molblock([("D", 0), ("O", 0), ("T", 0)], [(1, 2, 1), (2, 3, 1)])).graph(0)[0][:, [0, 2]].tolist() == [[1, 2], [8, 0], [1, 3]]
    # radicals take the place of hydrogens: methyl (M  RAD doublet, or charge code 4), methylene (triplet), a nitroxide
    assert one(molblock([("C", 0)], [], ["M  RAD  1   1   2"])).graph(0)[0][:, 3].tolist() == [3]
    assert one(molblock([("C", 4)], [])).graph(0)[0][:, 3].tolist() == [3]
    assert one(molblock([("C", 0)], [], ["M  RAD  1   1   3"])).graph(0)[0][:, 3].tolist() == [2]
    assert one(molblock([("N", 0), ("O", 0), ("C", 0), ("C", 0)], [(1, 2, 1), (1, 3, 1), (1, 4, 1)],
                        ["M  RAD  1   2   2"])).graph(0)[0][:, 3].tolist() == [0, 0, 3, 3]
    # salts and metals: no implicit hydrogens where RDKit has no valence list
    assert one(molblock([("Na", 3), ("Cl", 5), ("Fe", 0)], [])).graph(0)[0][:, [0, 1, 3]].tolist() == [[11, 1, 0], [17, -1, 0], [26, 0, 0]]
    # RDKit's clean-up applies to molfiles as well: a five-valent nitro group
    assert one(molblock([("C", 0), ("N", 0), ("O", 0), ("O", 0)], [(1, 2, 1), (2, 3, 2), (2, 4, 2)])).graph(0)[0][:, 1].tolist() == [0, 1, -1, 0]
    # unsupported or broken records are refused one by one; the others of the file are unaffected
    bad = [molblock([("C", 0)], [], version="V3000"), molblock([("R#", 0)], []), molblock([("C", 0)], [], ["M  RAD  1   1   7"]),
           molblock([("C", 8)], []), molblock([("C", 0), ("C", 0)], [(1, 2, 5)]), molblock([("C", 0), ("C", 0)], [(1, 3, 1)]),
           molblock([("C", 0), ("C", 0)], [(1, 2, 1), (2, 1, 1)]), "too\nshort\n"]
    text = "$$$$\n".join(bad + [benzene]) + "$$$$\n"
    s = SmilesSet.from_sdf_text(text)
    assert s.status.tolist() == [1] * len(bad) + [0]
    # a record without M  END, a file without a final $$$$, DOS line ends, data items after the molfile, an empty file
    assert one(molblock([("C", 0)], [], end=False)).graph(0)[0][:, 3].tolist() == [4]
    s = SmilesSet.from_sdf_text((benzene + "> <name>\nbenzene\n\n$$$$\n" + kekule).replace("\n", "\r\n"))
    assert s.status.tolist() == [0, 0] and s.n_atoms.tolist() == [6, 6]
    assert len(SmilesSet.from_sdf_text("")) == 0 and len(SmilesSet.from_sdf_text("\n\n")) == 0
    with pytest.raises(ValueError, match="not ingested"):
        SmilesSet.from_sdf_text(bad[0] + "$$$$\n").morgan_inputs([0], 32)


def test_threads_and_chunks_keep_record_order():
    text = (GOLDEN / FILES[0]).read_text() * 4                                 # 764 records: more than one chunk of 512
    a, b = SmilesSet.from_sdf_text(text, 1), SmilesSet.from_sdf_text(text, 8)
    assert len(a) == 764 and np.array_equal(a.n_atoms, b.n_atoms) and np.array_equal(a.status, b.status)
    assert np.array_equal(a.n_atoms[:191], a.n_atoms[191:382]) and np.array_equal(a.graph(5)[0], b.graph(5 + 573)[0])


def test_chembl_molecules_written_as_hydrogen_free_kekule_molfiles_come_back_as_rdkit_wrote_them():
    """The other direction: molfiles that draw NO hydrogen.  Every molecule of the reference's benchmark SMILES (written by
    RDKit, so the hydrogen counts in its bracket atoms are RDKit's) is turned into a Kekule-form molfile without hydrogens by
    the oracle; the SD entry has to find all hydrogen counts with the valence model ([NH3+], [O-], [nH] from its Kekule
    neighbourhood, [n+], [S+], ...), perceive the aromaticity again and arrive at the graph of the SMILES entry."""
    from oracle import smiles as osmi

    smiles = [line.split()[0] for line in (GOLDEN / "chembl_10k.smi").read_text().splitlines() if line.strip()]
    blocks, tables = [], []
    for smi in smiles:
        atoms, bonds = osmi.molecule(smi)
        types = oarom.kekulize(atoms, bonds) if (bonds[:, 2] == 12).any() else bonds[:, 2]
        blocks.append(omol.write_molblock(atoms, types, bonds))
        tables.append((atoms, bonds))
    got = SmilesSet.from_sdf_text("".join(blocks))
    assert np.all(got.status == 0)
    different = []
    for i, (atoms, bonds) in enumerate(tables):
        ga, gb = got.graph(i)
        if not (np.array_equal(ga, atoms) and np.array_equal(gb, bonds)):
            different.append(i)
    assert not different, [smiles[i] for i in different]                 # (one of them is a nitroxide radical: M  RAD)
    assert sum(int((a[:, 1] != 0).any()) for a, _ in tables) > 500      # charged molecules took part
