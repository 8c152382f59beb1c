"""Morgan fingerprints against values RDKit itself publishes (the algorithm is RDKit's; the reference asserts bit-for-bit
equality with it, tests/test_morgan_fingerprint.cpp:73-108, but holds no literal values).  The numbers below are from the RDKit
documentation, "Getting Started with the RDKit in Python":
  * "Explaining bits from Morgan Fingerprints": m = MolFromSmiles('c1cccnc1C'), radius 2 -> 16 non-zero elements;
    info[98513984] == ((1, 1), (2, 1)) (two atoms, radius 1) and info[4048591891] == ((5, 2),) (one atom, radius 2);
  * "Generating images of fingerprint bits": mol = MolFromSmiles('c1ccccc1CC1CC1'), GetMorganFingerprintAsBitVect(mol, radius=2)
    (2048 bits): bi[872] == ((6, 2),) — and bit 29 heads list(fp.GetOnBits());
  * the radius-0 identifiers every RDKit user has seen in GetNonzeroElements(): 2246728737 (CH3), 2245384272 (CH2),
    864662311 (OH), 847957139 (NH2), 3218693969 (aromatic CH).
  * "Morgan Fingerprints (Circular Fingerprints)": m1 = 'Cc1ccccc1', m2 = 'Cc1ncccc1', radius 2: DiceSimilarity of the count
    fingerprints prints 0.55..., of the 1024-bit vectors 0.51... (= 14 / 27: 11 and 16 bits set, 7 in common, which makes the
    Tanimoto similarity 7 / 20).
They go through the whole chain — the library's SMILES ingestion, its atom / bond invariants, then the oracle's environments
(CPU) and the HIP kernel's bits (GPU) — so they pin the invariant recipe and the hash chain to RDKit, not to a restatement."""

import numpy as np
import pytest

import oracle
from nvmolkit_amd.fingerprints import MorganFingerprintGenerator, SmilesSet


def environments(smiles, radius):
    s = SmilesSet([smiles])
    assert s.status[0] == 0
    atom_inv, bond_inv, bond_idx, bond_other, n_atoms = s.morgan_inputs([0], 32)
    return oracle.morgan_environments(atom_inv[0], bond_inv[0], bond_idx[0], bond_other[0], int(n_atoms[0]), radius)


def test_explaining_bits_example():
    codes, layers = environments("c1cccnc1C", 2)
    assert len(set(codes.tolist())) == 16
    assert [int(l) for c, l in zip(codes, layers) if int(c) == 98513984] == [1, 1]
    assert [int(l) for c, l in zip(codes, layers) if int(c) == 4048591891] == [2]


def test_bit_image_example():
    codes, layers = environments("c1ccccc1CC1CC1", 2)
    bits = sorted({int(c) % 2048 for c in codes})
    assert bits[0] == 29 and 872 in bits
    assert [int(l) for c, l in zip(codes, layers) if int(c) % 2048 == 872] == [2]


@pytest.mark.parametrize("smiles,want", [("CC", {2246728737}), ("CCO", {864662311, 2245384272, 2246728737}),
                                         ("CCN", {847957139, 2245384272, 2246728737}), ("c1ccccc1", {3218693969})])
def test_radius_zero_identifiers(smiles, want):
    codes, _ = environments(smiles, 0)
    assert set(codes.tolist()) == want


def test_count_fingerprints_everyone_has_seen():
    """GetMorganFingerprint(mol, 2).GetNonzeroElements() of benzene and ethanol as RDKit prints them (they turn up in the
    documentation, in the mailing list and in countless notebooks)."""
    from collections import Counter

    assert Counter(environments("c1ccccc1", 2)[0].tolist()) == {98513984: 6, 2763854213: 6, 3218693969: 6}
    assert Counter(environments("CCO", 2)[0].tolist()) == {864662311: 1, 1535166686: 1, 2245384272: 1, 2246728737: 1,
                                                            3542456614: 1, 4018048386: 1}


def test_dice_similarity_example():
    from collections import Counter

    counts = [Counter(environments(smi, 2)[0].tolist()) for smi in ("Cc1ccccc1", "Cc1ncccc1")]
    common = sum(min(counts[0][k], counts[1][k]) for k in counts[0])
    assert 2 * common / (sum(counts[0].values()) + sum(counts[1].values())) == pytest.approx(0.55, abs=1e-12)   # docs: 0.55...
    s = SmilesSet(["Cc1ccccc1", "Cc1ncccc1"])
    fp = oracle.morgan_fingerprints(*s.morgan_inputs([0, 1], 32), 32, 2, 1024)
    a, b, c = (int(np.unpackbits(x.view(np.uint8)).sum()) for x in (fp[0], fp[1], fp[0] & fp[1]))
    assert (a, b, c) == (11, 16, 7) and str(2 * c / (a + b)).startswith("0.51")                                  # docs: 0.51...
