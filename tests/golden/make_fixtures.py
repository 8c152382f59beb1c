"""Generates the committed fixtures under tests/golden/.

Run from the repository root:  python tests/golden/make_fixtures.py

Sources of the EXPECTED values:
  * similarity_handcomputed.npz — computed below with exact rational arithmetic
    (python `fractions`) from explicit bit sets; independent of both the oracle and the product.
  * butina_10x10.npz — the reference's known-answer case: distance matrix and expected clustering
    are DATA held by the reference test tests/test_butina.cpp:241-273 (matrix literal and the
    EXPECT_* values); nothing is executed from the reference.
  * fingerprints_*.npz — seeded numpy inputs (no expected outputs); the reference's tests draw the
    equivalent inputs from torch's RNG (nvmolkit/tests/test_clustering.py:154-163), which is not
    stable across torch versions, hence the committed arrays.
The reference itself cannot be executed in the build container (no RDKit, no CUDA), see DESIGN.md.
"""

from __future__ import annotations

import sys
from fractions import Fraction
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parents[1]))
from tests.util import clustered_fingerprints, pack_bits, random_fingerprints  # noqa: E402


def handcomputed():
    # 6 fingerprints of 128 bits given as explicit on-bit sets
    sets = [
        set(),                                   # all zero
        {0},                                     # single bit, word 0 bit 0
        {0, 31, 32, 63, 64, 95, 96, 127},        # word boundaries
        set(range(0, 128, 2)),                   # even bits
        set(range(0, 128)),                      # all ones
        {1, 2, 3, 5, 8, 13, 21, 34, 55, 89},     # fibonacci positions
    ]
    bits = np.zeros((len(sets), 128), dtype=bool)
    for i, s in enumerate(sets):
        bits[i, sorted(s)] = True
    n = len(sets)
    tani = np.zeros((n, n))
    cos = np.zeros((n, n))
    inter = np.zeros((n, n), dtype=np.int32)
    for i in range(n):
        for j in range(n):
            c = len(sets[i] & sets[j])
            u = len(sets[i] | sets[j])
            inter[i, j] = c
            tani[i, j] = float(Fraction(c, max(u, 1)))  # exact rational -> nearest double == IEEE c/u
            pa, pb = len(sets[i]), len(sets[j])
            cos[i, j] = 0.0 if c == 0 or pa * pb == 0 else c / np.sqrt(float(pa) * float(pb))
    np.savez(HERE / "similarity_handcomputed.npz", words=pack_bits(bits), tanimoto=tani, cosine=cos,
             intersection=inter)


def butina_known_answer():
    # tests/test_butina.cpp:248-254 (10x10 matrix), cutoff 0.1 (:243); expected :258-272
    rows = [
        [0.0, 0.05, 0.05, 0.05, 1, 1, 1, 1, 1, 1],
        [0.05, 0.0, 1, 1, 1, 1, 1, 1, 1, 1],
        [0.05, 1, 0.0, 1, 1, 1, 1, 1, 1, 1],
        [0.05, 1, 1, 0.0, 1, 1, 1, 1, 1, 1],
        [1, 1, 1, 1, 0.0, 0.05, 0.05, 1, 1, 1],
        [1, 1, 1, 1, 0.05, 0.0, 1, 1, 1, 1],
        [1, 1, 1, 1, 0.05, 1, 0.0, 1, 1, 1],
        [1, 1, 1, 1, 1, 1, 1, 0.0, 1, 1],
        [1, 1, 1, 1, 1, 1, 1, 1, 0.0, 1],
        [1, 1, 1, 1, 1, 1, 1, 1, 1, 0.0],
    ]
    np.savez(HERE / "butina_10x10.npz", dist=np.array(rows, dtype=np.float64), cutoff=0.1, n_clusters=5,
             cluster0=np.array([0, 1, 2, 3]), centroid0=0, cluster1=np.array([4, 5, 6]), centroid1=4)


def fingerprint_sets():
    np.savez_compressed(HERE / "fingerprints_small.npz",
                        random_128x64=random_fingerprints(128, 64, density=0.05),
                        random_77x4=random_fingerprints(77, 4, density=0.3),
                        clustered_300x32=clustered_fingerprints(300, 32, 12, max_flips=20, density=0.1),
                        clustered_500x64=clustered_fingerprints(500, 64, 25, max_flips=12, density=0.023))


if __name__ == "__main__":
    handcomputed()
    butina_known_answer()
    fingerprint_sets()
    print("fixtures written to", HERE)
