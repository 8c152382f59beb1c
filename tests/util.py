"""Shared helpers for the test-suite: seeded synthetic fingerprints and the Butina property checkers
(modelled on the reference's own checkers: tests/test_butina.cpp:96-153 and
nvmolkit/tests/test_clustering.py:23-51)."""

from __future__ import annotations

import numpy as np

SEED = 20260926


def random_fingerprints(n: int, words: int, density: float = 0.05, seed: int = SEED) -> np.ndarray:
    """(n, words) uint32 with i.i.d. bits at the given density."""
    rng = np.random.default_rng(seed)
    bits = rng.random((n, words * 32)) < density
    return pack_bits(bits)


def clustered_fingerprints(n: int, words: int, n_centres: int, max_flips: int = 12, density: float = 0.023,
                           seed: int = SEED, shuffle: bool = True) -> np.ndarray:
    """Planted clusters: each row = one of n_centres random centres with up to max_flips bit flips
    (the BASELINE.md cfg2 generator, scaled down)."""
    rng = np.random.default_rng(seed)
    nbits = words * 32
    centres = rng.random((n_centres, nbits)) < density
    owner = np.arange(n) % n_centres
    bits = centres[owner].copy()
    flips = rng.integers(0, max_flips + 1, size=n)
    for i in range(n):
        if flips[i]:
            pos = rng.integers(0, nbits, size=flips[i])
            bits[i, pos] ^= True
    if shuffle:
        bits = bits[rng.permutation(n)]
    return pack_bits(bits)


def pack_bits(bits: np.ndarray) -> np.ndarray:
    n, nb = bits.shape
    assert nb % 32 == 0
    b = bits.reshape(n, nb // 32, 32).astype(np.uint32)
    return (b << np.arange(32, dtype=np.uint32)).sum(axis=2, dtype=np.uint32)


def check_partition(clusters, n):
    seen = [m for c in clusters for m in c]
    assert sorted(seen) == list(range(n)), "every row must be assigned exactly once"


def check_greedy_butina(hit: np.ndarray, clusters) -> None:
    """Greedy-max property: each non-singleton cluster has exactly the largest neighbourhood that was
    still available when it was taken; after the first singleton only singletons follow."""
    hit = hit.copy()
    n = hit.shape[0]
    seen = set()
    for i, cl in enumerate(clusters):
        assert len(cl) > 0
        if len(cl) == 1:
            rest = [c for c in clusters[i:]]
            assert all(len(c) == 1 for c in rest), "non-singleton after a singleton"
            items = [c[0] for c in rest]
            assert len(set(items)) == len(items) and seen.isdisjoint(items)
            seen.update(items)
            break
        counts = hit.sum(axis=1)
        assert len(cl) == counts.max(), f"cluster {i}: size {len(cl)} != best available {counts.max()}"
        for m in cl:
            assert m not in seen
            seen.add(m)
        idx = list(cl)
        hit[idx, :] = False
        hit[:, idx] = False
    assert len(seen) == n


def check_labels_valid(hit: np.ndarray, labels: np.ndarray) -> None:
    """The C++ reference checker: sizes non-increasing by id, partition, and every cluster has a
    member adjacent to all the others (tests/test_butina.cpp:96-153)."""
    n = len(labels)
    k = int(labels.max()) + 1 if n else 0
    clusters = [np.flatnonzero(labels == c) for c in range(k)]
    sizes = [len(c) for c in clusters]
    assert all(s > 0 for s in sizes)
    assert all(sizes[i] >= sizes[i + 1] for i in range(k - 1)), "cluster ids not ordered by size"
    assert sum(sizes) == n
    for c in clusters:
        ok = any(all(hit[cen, m] for m in c if m != cen) for cen in c)
        assert ok, "cluster without a valid centroid"
