"""synthetic.graph_molecule: flattened ETKDG / MMFF tables from a REAL molecular graph (the library's SMILES ingestion) with
generic parameters — what the conformer benchmark runs on the reference's own molecules (benchmarks/etkdg_bench.py:154-161,
benchmarks/bench_utils/molprep.py:21-55: AddHs) while RDKit's parameter tables are out of reach.  No GPU."""

from pathlib import Path

import numpy as np
import pytest

from nvmolkit_amd import synthetic
from nvmolkit_amd.fingerprints import SmilesSet
from nvmolkit_amd.forcefield import ETK, GROUP_LAYOUT, MMFF, DG

GOLDEN = Path(__file__).parent / "golden"


def build(smiles, seed=0, **kw):
    s = SmilesSet([smiles])
    assert s.status[0] == 0
    return synthetic.graph_molecule(*s.graph(0), np.random.default_rng(seed), **kw)


def bounds_matrix(m):
    n = m["embed"]["n_atoms"]
    pairs, lb, ub = m["bounds"]
    L, U = np.zeros((n, n)), np.zeros((n, n))
    L[pairs[:, 0], pairs[:, 1]] = L[pairs[:, 1], pairs[:, 0]] = lb
    U[pairs[:, 0], pairs[:, 1]] = U[pairs[:, 1], pairs[:, 0]] = ub
    return L, U


def test_hydrogens_are_added_from_the_valence_model():
    for smi, n_atoms, n_h in [("CCO", 9, 6), ("c1ccccc1", 12, 6), ("CC(=O)N", 9, 5), ("[NH4+]", 5, 4), ("C#N", 3, 1), ("OC(=O)C(F)(F)F", 8, 1)]:
        m = build(smi)
        assert m["embed"]["n_atoms"] == n_atoms and int((m["elements"] == 1).sum()) == n_h, smi
        assert len(m["bonds"]) == n_atoms - 1 + (1 if "1" in smi else 0)


def test_benzene_bounds_are_the_planar_hexagon():
    m = build("c1ccccc1")
    L, U = bounds_matrix(m)
    cc = 2 * 0.76 - 0.12
    for a in range(6):
        for k, want in ((1, cc), (2, cc * np.sqrt(3.0)), (3, 2.0 * cc)):   # ortho, meta, para
            b = (a + k) % 6
            assert L[a, b] - 0.07 <= want <= U[a, b] + 0.07 and U[a, b] - L[a, b] < 0.2, (a, b, L[a, b], U[a, b], want)
    # every carbon is a planar centre: one improper (three permutations) and one out-of-plane term set each
    assert m["embed"]["num_impropers"] == 6 and len(m["embed"]["etk"][1][0]) == 18 and len(m["mmff"][3][0]) == 18
    assert len(m["embed"]["etk"][0][0]) == 0                              # no rotatable bond, no torsion preference


def test_bounds_are_consistent_windows_that_obey_the_triangle_inequality():
    s = SmilesSet.from_file(str(GOLDEN / "chembl_1k.smi"))
    rng = np.random.default_rng(1)
    done = 0
    for i in range(0, 400, 7):
        m = synthetic.graph_molecule(*s.graph(i), rng, max_atoms=90)
        if m is None:
            continue
        done += 1
        L, U = bounds_matrix(m)
        n = len(L)
        off = ~np.eye(n, dtype=bool)
        assert (L[off] > 0).all() and (L[off] < U[off]).all()
        assert (U <= (U[:, :, None] + U[None, :, :].transpose(0, 2, 1)).min(1) + 1e-9).all()      # u_ij <= u_ik + u_kj
        for a, b in m["bonds"]:
            assert U[a, b] - L[a, b] == pytest.approx(0.02, abs=1e-9) and 0.9 < L[a, b] < 2.2
    assert done >= 30


def test_tables_have_the_flattened_layout_and_valid_indices():
    m = build("CC(C)Cc1ccc(cc1)[C@@H](C)C(=O)O", seed=3)                   # ibuprofen
    n = m["embed"]["n_atoms"]
    assert n == 33
    for kind, groups in ((DG, m["embed"]["dg"]), (ETK, m["embed"]["etk"]), (MMFF, m["mmff"])):
        for (n_idx, n_par), (idx, par) in zip(GROUP_LAYOUT[kind], groups):
            idx, par = np.asarray(idx), np.asarray(par)
            assert idx.shape == (len(idx), n_idx) and par.shape == (len(idx), n_par)
            assert len(idx) == 0 or (idx.min() >= 0 and idx.max() < n)
    assert len(m["embed"]["dg"][0][0]) == n * (n - 1) // 2
    # the carboxylic acid and the ring: planar centres; the isobutyl / alpha carbons: torsion preferences about rotatable bonds
    assert m["embed"]["num_impropers"] == 7 and len(m["embed"]["etk"][0][0]) >= 4
    kinds = [c[0] for c in m["embed"]["checks"]]
    assert 0 not in kinds                                                  # no centre shared by two rings: no tetrahedral-shape check


def test_rings_fused_centres_and_fragments():
    dec = build("C1CCC2CCCCC2C1")                                          # decalin: two tetrahedral centres shared by two rings
    assert sum(1 for c in dec["embed"]["checks"] if c[0] == 0) == 2
    salt = build("CC(=O)[O-].[Na+]")                                       # two fragments: far apart in the graph, bounded by the floors only
    L, U = bounds_matrix(salt)
    na = int(np.where(salt["elements"] == 11)[0][0])
    assert (U[na, np.arange(len(U)) != na] >= 999.0).all()
    cyc = build("C1CC1C")                                                  # three-membered ring: 60 degree angles, bonded 1-3 pairs
    L, U = bounds_matrix(cyc)
    assert U[0, 2] < 1.6


def test_library_is_deterministic_and_filters_by_size():
    a, ids_a = synthetic.smiles_file_library(GOLDEN / "chembl_1k.smi", n_mols=60, max_atoms=64, processes=1)
    b, ids_b = synthetic.smiles_file_library(GOLDEN / "chembl_1k.smi", n_mols=60, max_atoms=64, processes=2)
    assert ids_a == ids_b and 10 < len(a) < 60 and all(m["embed"]["n_atoms"] <= 64 for m in a)
    for x, y in zip(a, b):
        assert np.array_equal(x["bounds"][1], y["bounds"][1]) and np.array_equal(x["mmff"][4][1], y["mmff"][4][1])
