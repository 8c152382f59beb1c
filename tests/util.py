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


# ---- flattened molecular graphs for the Morgan path ------------------------------------------------
# Layout of MorganInvariantsGenerator::ComputeInvariantsInto (reference src/morgan_fingerprint_common.cpp:43-124).

BOND_SINGLE, BOND_DOUBLE, BOND_TRIPLE, BOND_AROMATIC = 1, 2, 3, 12  # RDKit::Bond::BondType values
MAX_BONDS_PER_ATOM = 8


def atom_invariant(z: int, heavy_degree: int, n_h: int, charge: int = 0, delta_mass: int = 0, in_ring: bool = False,
                   neighbor_h: int = 0) -> int:
    """[Z, degree + Hs, total Hs incl. H neighbours, formal charge, int(mass - avg mass)] (+ [1] if in a ring),
    hashed with hash_range on uint32 (src/morgan_fingerprint_common.cpp:96-121)."""
    import oracle

    comps = [z, heavy_degree + n_h, n_h + neighbor_h, charge & 0xFFFFFFFF, delta_mass & 0xFFFFFFFF]
    if in_ring:
        comps.append(1)
    return oracle.morgan_hash_vector(comps)


def flatten_molecules(mols, stride: int):
    """mols: list of (atoms, bonds); atoms = [(Z, nH, charge, in_ring)], bonds = [(i, j, type)].
    Returns (atom_inv, bond_inv, bond_idx, bond_other, n_atoms) numpy arrays in the reference layout."""
    n = len(mols)
    atom_inv = np.zeros((n, stride), dtype=np.uint32)
    bond_inv = np.zeros((n, stride), dtype=np.uint32)
    bond_idx = np.full((n, stride, MAX_BONDS_PER_ATOM), -1, dtype=np.int16)
    bond_other = np.full((n, stride, MAX_BONDS_PER_ATOM), -1, dtype=np.int16)
    n_atoms = np.zeros(n, dtype=np.int16)
    for m, (atoms, bonds) in enumerate(mols):
        assert len(atoms) < stride and len(bonds) < stride, "reference buckets require atoms, bonds < maxAtoms"
        n_atoms[m] = len(atoms)
        deg = [0] * len(atoms)
        for b, (i, j, t) in enumerate(bonds):
            bond_inv[m, b] = t
            for a, o in ((i, j), (j, i)):
                assert deg[a] < MAX_BONDS_PER_ATOM
                bond_idx[m, a, deg[a]] = b
                bond_other[m, a, deg[a]] = o
                deg[a] += 1
        for a, (z, nh, q, ring) in enumerate(atoms):
            atom_inv[m, a] = atom_invariant(z, deg[a], nh, q, 0, ring)
    return atom_inv, bond_inv, bond_idx, bond_other, n_atoms


def random_molecule(rng, n_atoms: int, ring_closures: int = 2, symmetric: bool = False):
    """Random connected graph with valence <= 4: a random tree plus a few ring closures.  `symmetric`
    draws few distinct atom types so that duplicate environments (the dedup logic) are exercised."""
    zs = [6, 6, 6, 7, 8] if symmetric else [6, 6, 6, 6, 7, 8, 9, 16, 17]
    deg = [0] * n_atoms
    bonds = []
    for a in range(1, n_atoms):
        cand = [p for p in range(a) if deg[p] < 3]
        p = cand[rng.integers(len(cand))] if cand else int(np.argmin(deg[:a]))
        bonds.append((p, a, BOND_SINGLE if rng.random() < 0.8 else BOND_DOUBLE))
        deg[p] += 1
        deg[a] += 1
    ring = [False] * n_atoms
    existing = {(min(i, j), max(i, j)) for i, j, _ in bonds}
    for _ in range(ring_closures):
        i, j = sorted(rng.integers(0, n_atoms, size=2).tolist())
        if i != j and (i, j) not in existing and deg[i] < 4 and deg[j] < 4 and len(bonds) < n_atoms - 1 + ring_closures:
            bonds.append((i, j, BOND_AROMATIC if rng.random() < 0.5 else BOND_SINGLE))
            existing.add((i, j))
            deg[i] += 1
            deg[j] += 1
            ring[i] = ring[j] = True
    atoms = []
    for a in range(n_atoms):
        z = zs[rng.integers(len(zs))]
        atoms.append((z, max(0, 4 - deg[a] - (z - 6 if z in (7, 8, 9) else 0)) if z != 17 else 0, 0, ring[a]))
    return atoms, bonds


def random_molecule_batch(n_mols: int, stride: int, seed: int = SEED, min_atoms: int = 1, symmetric: bool = False):
    rng = np.random.default_rng(seed)
    mols = []
    for _ in range(n_mols):
        hi = stride - 3
        n = int(rng.integers(min_atoms, max(min_atoms + 1, hi)))
        mols.append(random_molecule(rng, n, ring_closures=min(2, max(0, stride - 1 - n)), symmetric=symmetric))
    return mols


# ---- synthetic flattened force-field systems / embeddable molecules (shared with the benchmarks) ------------------
from nvmolkit_amd.synthetic import (_chain_positions, build_ff_batch_arrays, random_ff_system,  # noqa: E402,F401
                                    synthetic_embed_molecule)
