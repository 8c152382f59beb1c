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


# ---- synthetic flattened force-field systems ---------------------------------------------------------

def _chain_positions(rng, n, dim):
    """Self-avoiding-ish random chain, ~1.5 A steps, so that no two atoms sit on top of each other."""
    pos = np.zeros((n, dim))
    for a in range(1, n):
        for _ in range(50):
            step = rng.normal(size=3)
            cand = pos[a - 1, :3] + 1.5 * step / np.linalg.norm(step)
            if a < 2 or np.min(np.linalg.norm(pos[:a - 1, :3] - cand, axis=1)) > 1.1:
                break
        pos[a, :3] = cand
    if dim == 4:
        pos[:, 3] = rng.normal(scale=0.3, size=n)
    return pos


def random_ff_system(kind: int, n_atoms: int, rng):
    """(pos (n, dim), groups [(idx, par)]) for one synthetic system of the given force-field kind.
    Parameters are drawn from the ranges of real tables; geometry terms straddle their bounds so that both the
    zero and the non-zero branches are exercised."""
    from oracle import ff as off

    dim = off.DIM[kind]
    pos = _chain_positions(rng, n_atoms, dim)
    n = n_atoms
    pairs = np.array([(i, j) for i in range(n) for j in range(i + 1, n)], dtype=np.int64).reshape(-1, 2)
    d = np.sqrt(((pos[pairs[:, 0], :3] - pos[pairs[:, 1], :3]) ** 2).sum(1)) if len(pairs) else np.zeros(0)
    chain4 = np.array([(i, i + 1, i + 2, i + 3) for i in range(n - 3)], dtype=np.int64).reshape(-1, 4)
    chain3 = np.array([(i, i + 1, i + 2) for i in range(n - 2)], dtype=np.int64).reshape(-1, 3)
    chain2 = np.array([(i, i + 1) for i in range(n - 1)], dtype=np.int64).reshape(-1, 2)
    far = pairs[(pairs[:, 1] - pairs[:, 0]) >= 3] if len(pairs) else pairs

    def quads(k):
        if n < 4:
            return np.zeros((0, 4), dtype=np.int64)
        return np.array([rng.choice(n, size=4, replace=False) for _ in range(k)], dtype=np.int64)

    if kind == off.QUARTIC:
        return pos, []
    if kind == off.DG:
        lb = d * rng.uniform(0.7, 1.2, size=len(d))
        ub = lb * rng.uniform(1.0, 1.3, size=len(d))
        q = quads(max(1, n // 4))
        vol = np.array([np.dot(pos[a, :3] - pos[dd, :3], np.cross(pos[b, :3] - pos[dd, :3], pos[c, :3] - pos[dd, :3]))
                        for a, b, c, dd in q]).reshape(-1)
        lo = vol + rng.uniform(-2.0, 1.0, size=len(q))
        return pos, [(pairs, np.stack([lb**2, ub**2, rng.uniform(0.5, 2.0, size=len(d))], 1) if len(d) else np.zeros((0, 3))),
                     (q, np.stack([lo, lo + rng.uniform(0.1, 1.5, size=len(q))], 1) if len(q) else np.zeros((0, 2))),
                     (np.arange(n, dtype=np.int64).reshape(-1, 1), np.zeros((n, 0)))]
    if kind == off.ETK:
        tors_par = np.concatenate([rng.uniform(0.0, 4.0, size=(len(chain4), 6)), rng.choice([-1.0, 1.0], size=(len(chain4), 6))], 1)
        imp = quads(max(1, n // 5))
        imp_par = np.stack([rng.uniform(0, 1, len(imp)), rng.uniform(-1, 1, len(imp)), rng.uniform(0, 0.5, len(imp)),
                            rng.uniform(1, 10, len(imp))], 1) if len(imp) else np.zeros((0, 4))

        def flat_bottom(ix, k):
            dd = np.sqrt(((pos[ix[:, 0], :3] - pos[ix[:, 1], :3]) ** 2).sum(1)) if len(ix) else np.zeros(0)
            lo_ = dd * rng.uniform(0.8, 1.15, size=len(dd))
            return np.stack([lo_, lo_ * rng.uniform(1.0, 1.2, size=len(dd)), np.full(len(dd), k), np.zeros(len(dd))], 1) if len(dd) else np.zeros((0, 4))

        d13 = np.array([(i, i + 2) for i in range(n - 2)], dtype=np.int64).reshape(-1, 2)
        ang_lo = rng.uniform(60, 130, size=len(chain3))
        return pos, [(chain4, tors_par), (imp, imp_par), (chain2, flat_bottom(chain2, 100.0)), (d13, flat_bottom(d13, 100.0)),
                     (chain3, np.stack([ang_lo, ang_lo + rng.uniform(0, 40, size=len(chain3))], 1) if len(chain3) else np.zeros((0, 2))),
                     (far, flat_bottom(far, 10.0))]
    if kind == off.UFF:
        # UFF (reference src/forcefields/uff.h:27-67): every angle order 0..4, torsion orders 2 / 3 / 6, inversions with
        # C2 = 0 (C, N, O centres) and C2 != 0 (group-15 centres), vdW cutoffs that exclude some of the pairs
        bond_par = np.stack([rng.uniform(1.0, 1.6, len(chain2)), rng.uniform(300.0, 900.0, len(chain2))], 1) if len(chain2) else np.zeros((0, 2))
        order = rng.integers(0, 5, size=len(chain3)).astype(float)
        th0 = np.deg2rad(rng.uniform(95, 125, len(chain3)))
        s0, c0 = np.sin(th0), np.cos(th0)
        c2 = 1.0 / (4.0 * np.maximum(s0 * s0, 1e-8))
        ang_par = np.stack([th0, rng.uniform(50.0, 200.0, len(chain3)), order, c2 * (2.0 * c0 * c0 + 1.0), -4.0 * c2 * c0, c2], 1) \
            if len(chain3) else np.zeros((0, 6))
        tors_par = np.stack([rng.uniform(0.5, 10.0, len(chain4)), rng.choice([2.0, 3.0, 6.0], size=len(chain4)),
                             rng.choice([-1.0, 1.0], size=len(chain4))], 1) if len(chain4) else np.zeros((0, 3))
        inv = quads(max(2, n // 4))
        grp15 = rng.random(len(inv)) < 0.5
        w0 = np.deg2rad(rng.uniform(80, 95, len(inv)))
        inv_par = np.stack([rng.uniform(2.0, 25.0, len(inv)), np.where(grp15, 4.0 * np.cos(w0) ** 2 - np.cos(2 * w0), 1.0),
                            np.where(grp15, -4.0 * np.cos(w0), -1.0), np.where(grp15, 1.0, 0.0)], 1) if len(inv) else np.zeros((0, 4))
        dfar = np.sqrt(((pos[far[:, 0], :3] - pos[far[:, 1], :3]) ** 2).sum(1)) if len(far) else np.zeros(0)
        xij = rng.uniform(3.0, 4.2, len(far))
        thr = np.where(rng.random(len(far)) < 0.25, dfar * 0.9, xij * 10.0)  # a quarter of the pairs sit beyond their cutoff
        vdw_par = np.stack([xij, rng.uniform(0.02, 0.3, len(far)), thr], 1) if len(far) else np.zeros((0, 3))
        return pos, [(chain2, bond_par), (chain3, ang_par), (chain4, tors_par), (inv, inv_par), (far, vdw_par)]
    # MMFF
    bond_par = np.stack([rng.uniform(1.0, 1.6, len(chain2)), rng.uniform(3.0, 8.0, len(chain2))], 1) if len(chain2) else np.zeros((0, 2))
    ang_par = np.stack([rng.uniform(100, 125, len(chain3)), rng.uniform(0.4, 1.2, len(chain3)),
                        (rng.random(len(chain3)) < 0.1).astype(float)], 1) if len(chain3) else np.zeros((0, 3))
    sb_par = np.stack([rng.uniform(100, 125, len(chain3)), rng.uniform(1.0, 1.6, len(chain3)), rng.uniform(1.0, 1.6, len(chain3)),
                       rng.uniform(-0.5, 0.5, len(chain3)), rng.uniform(-0.5, 0.5, len(chain3))], 1) if len(chain3) else np.zeros((0, 5))
    oop = quads(max(1, n // 5))
    vdw_par = np.stack([rng.uniform(3.0, 4.2, len(far)), rng.uniform(0.02, 0.2, len(far))], 1) if len(far) else np.zeros((0, 2))
    q = rng.uniform(-0.6, 0.6, size=n)
    ele_par = np.stack([q[far[:, 0]] * q[far[:, 1]], rng.choice([1.0, 2.0], size=len(far)),
                        ((far[:, 1] - far[:, 0]) == 3).astype(float)], 1) if len(far) else np.zeros((0, 3))
    return pos, [(chain2, bond_par), (chain3, ang_par), (chain3, sb_par), (oop, rng.uniform(0.01, 0.2, size=(len(oop), 1))),
                 (chain4, rng.uniform(-2.0, 2.0, size=(len(chain4), 3))), (far, vdw_par), (far, ele_par)]


def build_ff_batch_arrays(kind: int, systems):
    """systems: list of (pos, groups) -> (atom_starts, flat positions, [(starts, idx, par)]) for FlatForcefieldBatch."""
    from oracle import ff as off

    layout = off.LAYOUT[kind]
    atom_starts = np.zeros(len(systems) + 1, dtype=np.int32)
    for s, (pos, _) in enumerate(systems):
        atom_starts[s + 1] = atom_starts[s] + len(pos)
    flat = np.concatenate([p.reshape(-1) for p, _ in systems]) if systems else np.zeros(0)
    groups = []
    for g, (n_idx, n_par) in enumerate(layout):
        starts = np.zeros(len(systems) + 1, dtype=np.int32)
        idx_all, par_all = [], []
        for s, (_, gs) in enumerate(systems):
            idx, par = gs[g]
            starts[s + 1] = starts[s] + len(idx)
            idx_all.append(np.asarray(idx, dtype=np.int32).reshape(-1, n_idx))
            par_all.append(np.asarray(par, dtype=np.float64).reshape(len(idx), n_par))
        groups.append((starts, np.concatenate(idx_all) if idx_all else np.zeros((0, n_idx), np.int32),
                       np.concatenate(par_all) if par_all else np.zeros((0, n_par))))
    return atom_starts, flat, groups


# ---- synthetic molecules for the ETKDG pipeline -------------------------------------------------------

def synthetic_embed_molecule(rng, n_atoms: int, with_etk: bool = True):
    """A chain molecule whose distance bounds are derived from a hidden reference geometry, so a feasible
    embedding exists (up to mirror image).  Returns (FlatMolecule fields dict, reference coordinates)."""
    ref = _chain_positions(rng, n_atoms, 3)
    n = n_atoms
    pairs = np.array([(i, j) for i in range(n) for j in range(i + 1, n)], dtype=np.int64).reshape(-1, 2)
    d = np.sqrt(((ref[pairs[:, 0]] - ref[pairs[:, 1]]) ** 2).sum(1)) if len(pairs) else np.zeros(0)
    sep = (pairs[:, 1] - pairs[:, 0]) if len(pairs) else np.zeros(0, dtype=np.int64)
    tol = np.minimum(0.02 * sep.astype(float) ** 2, 1.0)
    lb, ub = np.maximum(d - tol, 0.5), d + tol
    dg = [(pairs, np.stack([lb**2, ub**2, np.ones(len(d))], 1) if len(d) else np.zeros((0, 3))),
          (np.zeros((0, 4), np.int64), np.zeros((0, 2))),
          (np.arange(n, dtype=np.int64).reshape(-1, 1), np.zeros((n, 0)))]
    etk = None
    if with_etk:
        def fb(mask, k):
            return pairs[mask], np.stack([lb[mask], ub[mask], np.full(mask.sum(), k), np.zeros(mask.sum())], 1)

        chain3 = np.array([(i, i + 1, i + 2) for i in range(n - 2)], dtype=np.int64).reshape(-1, 3)
        ang = []
        for i, j, k in chain3:
            a, b = ref[i] - ref[j], ref[k] - ref[j]
            ang.append(np.degrees(np.arccos(np.clip(a @ b / np.linalg.norm(a) / np.linalg.norm(b), -1, 1))))
        ang = np.array(ang).reshape(-1)
        chain4 = np.array([(i, i + 1, i + 2, i + 3) for i in range(n - 3)], dtype=np.int64).reshape(-1, 4)
        tors = np.concatenate([rng.uniform(0.0, 0.2, size=(len(chain4), 6)), rng.choice([-1.0, 1.0], size=(len(chain4), 6))], 1)
        etk = [(chain4, tors), (np.zeros((0, 4), np.int64), np.zeros((0, 4))), fb(sep == 1, 100.0), fb(sep == 2, 100.0),
               (chain3, np.stack([ang - 5.0, ang + 5.0], 1) if len(ang) else np.zeros((0, 2))), fb(sep >= 3, 10.0)]
    checks = [(5, (i, i + 1, i + 2), ()) for i in range(n - 2)]  # NVMK_CHECK_DOUBLE_BOND_GEOMETRY: never linear here
    return dict(n_atoms=n, dg=dg, etk=etk, checks=checks, num_impropers=0), ref, (pairs, lb, ub)
