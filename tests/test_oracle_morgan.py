"""Pins the Morgan oracle (oracle/oracle_morgan.c) without RDKit: hash arithmetic by hand, environment
deduplication through the element-count known answers RDKit's own test-suite holds and the reference
repeats (tests/test_morgan_fingerprint_ref.cpp:44-60), on hand-flattened graphs."""

import numpy as np
import pytest

import oracle
from tests import util
from tests.util import BOND_DOUBLE as D
from tests.util import BOND_SINGLE as S

C3 = (6, 3, 0, False)   # CH3
C2 = (6, 2, 0, False)   # CH2
OH = (8, 1, 0, False)
PENTANE = ([C3, C2, C2, C2, C3], [(0, 1, S), (1, 2, S), (2, 3, S), (3, 4, S)])
# O=C(O)CC1CC1 : O=, C, OH, CH2, ring CH, ring CH2, ring CH2
ACID_A = ([(8, 0, 0, False), (6, 0, 0, False), OH, C2, (6, 1, 0, True), (6, 2, 0, True), (6, 2, 0, True)],
          [(0, 1, D), (1, 2, S), (1, 3, S), (3, 4, S), (4, 5, S), (5, 6, S), (6, 4, S)])
# OC(=O)CC1CC1 : same molecule, different atom order
ACID_B = ([OH, (6, 0, 0, False), (8, 0, 0, False), C2, (6, 1, 0, True), (6, 2, 0, True), (6, 2, 0, True)],
          [(0, 1, S), (1, 2, D), (1, 3, S), (3, 4, S), (4, 5, S), (5, 6, S), (6, 4, S)])
DIOL = ([OH, C2, C2, C2, C2, OH], [(0, 1, S), (1, 2, S), (2, 3, S), (3, 4, S), (4, 5, S)])  # OCCCCO


def envs(mol, radius):
    ai, bi, bx, bo, na = util.flatten_molecules([mol], 32)
    return oracle.morgan_environments(ai[0], bi[0], bx[0], bo[0], int(na[0]), radius)


def test_hash_combine_by_hand():
    def hc(seed, v):
        return (seed ^ ((v + 0x9E3779B9 + ((seed << 6) & 0xFFFFFFFF) + (seed >> 2)) & 0xFFFFFFFF)) & 0xFFFFFFFF

    seed = 0
    for v in [6, 4, 3, 0, 0]:
        seed = hc(seed, v)
    assert oracle.morgan_hash_vector([6, 4, 3, 0, 0]) == seed
    assert oracle.morgan_hash_vector([]) == 0
    assert oracle.morgan_hash_vector([0]) == 0x9E3779B9


@pytest.mark.parametrize("mol,want", [(PENTANE, [2, 5, 7, 7]), (ACID_A, [6, 12, 16, 17]), (ACID_B, [6, 12, 16, 17])])
def test_nonzero_element_counts_known_answers(mol, want):
    # tests/test_morgan_fingerprint_ref.cpp:53-58 ("Vals taken from testMorganFP() in testFingerprintGenerators.cpp")
    for radius, n_expected in enumerate(want):
        codes, _ = envs(mol, radius)
        assert len(set(codes.tolist())) == n_expected, f"radius {radius}"


def test_atom_order_invariance():
    for radius in range(4):
        a, _ = envs(ACID_A, radius)
        b, _ = envs(ACID_B, radius)
        assert sorted(a.tolist()) == sorted(b.tolist())


def test_symmetry_counts():
    # tests/test_morgan_fingerprint_ref.cpp:60-69: OCCCCO radius 2 -> 7 distinct ids, each seen 2 or 4 times
    codes, _ = envs(DIOL, 2)
    vals, counts = np.unique(codes, return_counts=True)
    assert len(vals) == 7
    assert set(counts.tolist()) <= {2, 4}


def test_layers_and_degree_zero_atoms():
    lone = ([(6, 4, 0, False)], [])  # methane: one atom, no bonds -> only the radius-0 environment
    codes, layers = envs(lone, 3)
    assert len(codes) == 1 and layers.tolist() == [0]
    codes, layers = envs(PENTANE, 2)
    assert layers.tolist().count(0) == 5 and max(layers) == 2


def test_folded_fingerprint_bits():
    ai, bi, bx, bo, na = util.flatten_molecules([PENTANE, ACID_A, DIOL], 32)
    for fp_bits in (128, 2048):
        fps = oracle.morgan_fingerprints(ai, bi, bx, bo, na, 32, 2, fp_bits)
        for m, mol in enumerate([PENTANE, ACID_A, DIOL]):
            codes, _ = envs(mol, 2)
            want = np.zeros(fp_bits, dtype=bool)
            want[np.unique(codes % fp_bits)] = True
            assert np.array_equal(oracle.unpack_bits(fps[m:m + 1])[0], want)


def test_random_batch_is_deterministic_and_radius_monotone():
    mols = util.random_molecule_batch(50, 64, seed=3, symmetric=True)
    flat = util.flatten_molecules(mols, 64)
    prev = None
    for radius in range(4):
        fps = oracle.morgan_fingerprints(*flat, 64, radius, 1024)
        assert np.array_equal(fps, oracle.morgan_fingerprints(*flat, 64, radius, 1024))
        if prev is not None:
            assert np.all((prev & fps) == prev), "bits only get added as the radius grows"
        prev = fps
