"""Synthetic, chemistry-free workloads for the force-field / conformer path: random flattened term tables with
parameters drawn from the ranges of real tables, and chain "molecules" whose distance bounds come from a hidden
reference geometry (so a feasible embedding exists).  Used by the tests (tests/util.py re-exports them), by
tools/bench_conformers.py and by bench.py's secondary conformer measurement — the stand-in for the RDKit-derived
inputs the reference's benchmarks use (benchmarks/etkdg_bench.py, ff_optimize_bench.py), which need RDKit."""

from __future__ import annotations

import numpy as np

from nvmolkit_amd.forcefield import DG, DIM, ETK, GROUP_LAYOUT, MMFF, QUARTIC, UFF


class _Layout:
    """The names the generators were written against (oracle.ff exposes the same constants)."""

    DG, ETK, MMFF, QUARTIC, UFF = DG, ETK, MMFF, QUARTIC, UFF
    DIM = DIM
    LAYOUT = GROUP_LAYOUT


off = _Layout

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


def _stereo_checks(checks):
    """The generators' check lists in array form (embedMolecules.StereoChecks: the table builder copies them without a tuple walk)."""
    from nvmolkit_amd.embedMolecules import StereoChecks

    return StereoChecks(checks)


def random_ff_system(kind: int, n_atoms: int, rng):
    """(pos (n, dim), groups [(idx, par)]) for one synthetic system of the given force-field kind.
    Parameters are drawn from the ranges of real tables; geometry terms straddle their bounds so that both the
    zero and the non-zero branches are exercised."""
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


# ---- drug-like synthetic molecules: graph + consistent 3-D geometry + every table derived from it -------------------
#
# The benchmark workload of BASELINE.json configs[2] / [3] is "10k drug-like SMILES": RDKit would perceive the chemistry
# and derive the ETKDG bounds / MMFF94 tables.  Without RDKit the tables come from a generated molecule instead: a
# skeleton of rings and chains with hydrogens on the free valences is grown atom by atom with ideal bond lengths and
# angles (so a clash-free 3-D geometry EXISTS), and the distance bounds (1-2, 1-3, 1-4 cis/trans windows, van der Waals
# floors, triangle smoothing), chiral sets, experimental-torsion / improper / restraint terms, stereo checks and the
# MMFF94-shaped tables (rest lengths and angles = the geometry's, force constants from the ranges of the real tables) are
# all derived from that one geometry.  Minima therefore exist and the minimisers converge the way they do on real input,
# instead of running to their iteration caps on contradictory random tables.

_BOND_HEAVY, _BOND_RING, _BOND_H = 1.50, 1.39, 1.09


def _perp(rng, u):
    v = rng.normal(size=3)
    v -= v.dot(u) * u
    n = np.linalg.norm(v)
    return v / n if n > 1e-8 else _perp(rng, u)


def _unit(v):
    return v / np.linalg.norm(v)


class _Grower:
    def __init__(self, rng, n_atoms):
        self.rng = rng
        self.n_max = n_atoms
        self.pos = np.zeros((n_atoms, 3))
        self.n = 0
        self.heavy = np.zeros(n_atoms, dtype=bool)
        self.cap = np.zeros(n_atoms, dtype=np.int64)      # valence cap: 4 sp3, 3 sp2 / ring, 2, 1
        self.ring = np.full(n_atoms, -1, dtype=np.int64)  # ring id or -1
        self.nbr = [[] for _ in range(n_atoms)]
        self.bonds = []
        self.n_rings = 0

    def clash(self, p, exclude, lim):
        if self.n == 0:
            return False
        d = np.linalg.norm(self.pos[:self.n] - p, axis=1)
        d[exclude] = 10.0
        return bool(d.min() < lim)

    def add(self, p, heavy, cap, ring=-1):
        i = self.n
        self.pos[i], self.heavy[i], self.cap[i], self.ring[i] = p, heavy, cap, ring
        self.n += 1
        return i

    def bond(self, a, b):
        self.nbr[a].append(b)
        self.nbr[b].append(a)
        self.bonds.append((min(a, b), max(a, b)))

    def direction(self, a):
        """Ideal direction of the next substituent of atom a (None = saturated geometry)."""
        rng, us = self.rng, [_unit(self.pos[b] - self.pos[a]) for b in self.nbr[a]]
        cap = self.cap[a]
        if not us:
            return _unit(rng.normal(size=3))
        if len(us) == 1:
            theta = np.deg2rad({4: 109.5, 3: 120.0, 2: 106.0}.get(int(cap), 109.5))
            return np.cos(theta) * us[0] + np.sin(theta) * _perp(rng, us[0])
        if len(us) == 2:
            if np.linalg.norm(us[0] + us[1]) < 1e-6:  # collinear neighbours: any perpendicular direction
                return _perp(rng, us[0])
            mid = -_unit(us[0] + us[1])
            if cap == 3:
                return mid
            side = _unit(np.cross(us[0], us[1])) * (1.0 if rng.random() < 0.5 else -1.0)
            return _unit(np.cos(np.deg2rad(54.75)) * mid + np.sin(np.deg2rad(54.75)) * side)
        tot = us[0] + us[1] + us[2]
        if np.linalg.norm(tot) < 0.2:  # planar centre whose cap was raised: go out of the plane
            return _unit(np.cross(us[0], us[1])) * (1.0 if rng.random() < 0.5 else -1.0)
        return -_unit(tot)

    def free(self, a):
        return self.cap[a] - len(self.nbr[a])


def _grow_heavy(g, rng, n_heavy, n_atoms):
    """Add rings / chain atoms until the skeleton has n_heavy heavy atoms (or no room is left)."""
    stall = 0
    fails: dict = {}

    def blocked(a):  # a valence that keeps clashing is closed, so that growth moves elsewhere
        fails[a] = fails.get(a, 0) + 1
        if fails[a] >= (2 if len(g.nbr[a]) >= 2 else 6):  # with two or more neighbours the direction has <= 2 choices
            g.cap[a] = len(g.nbr[a])

    while int(g.heavy[:g.n].sum()) < n_heavy and stall < 400 and g.n < n_atoms:
        open_atoms = [a for a in range(g.n) if g.heavy[a] and g.free(a) > 0]
        if not open_atoms:
            # every valence is used: open one more on a terminal / divalent atom so the chain can go on
            low = [a for a in range(g.n) if g.heavy[a] and g.cap[a] < 4 and g.ring[a] < 0]
            if not low:
                return
            g.cap[low[int(rng.integers(0, len(low)))]] += 1
            continue
        a = open_atoms[-1 - int(rng.integers(0, min(4, len(open_atoms))))]  # recent atoms first: chain-like growth
        d = g.direction(a)
        left = n_heavy - int(g.heavy[:g.n].sum())
        size = 6 if rng.random() < 0.8 else 5
        if left >= size and rng.random() < 0.3 and g.n + size <= n_atoms:
            rad = _BOND_RING / (2.0 * np.sin(np.pi / size))
            p0 = g.pos[a] + _BOND_HEAVY * d
            centre = p0 + rad * d
            w = _perp(rng, d)
            pts = [centre + rad * (np.cos(2 * np.pi * k / size) * (-d) + np.sin(2 * np.pi * k / size) * w) for k in range(size)]
            if any(g.clash(p, [a], 2.1) for p in pts):
                stall += 1
                blocked(a)
                continue
            ids = [g.add(p, True, 3, g.n_rings) for p in pts]
            g.n_rings += 1
            g.bond(a, ids[0])
            for k in range(size):
                g.bond(ids[k], ids[(k + 1) % size])
        else:
            p = g.pos[a] + _BOND_HEAVY * d
            if g.clash(p, [a], 2.1):
                stall += 1
                blocked(a)
                continue
            cap = int(rng.choice([4, 3, 2, 1], p=[0.5, 0.25, 0.15, 0.10]))
            g.bond(a, g.add(p, True, cap))


def _grow_hydrogens(g, n_atoms):
    """Hydrogens on the free valences, round-robin, until the atom budget is used or nothing fits."""
    progress = True
    while g.n < n_atoms and progress:
        progress = False
        for a in range(g.n):
            if g.n >= n_atoms:
                break
            if not g.heavy[a] or g.free(a) <= 0:
                continue
            placed = False
            for _try in range(6):
                p = g.pos[a] + _BOND_H * g.direction(a)
                if not g.clash(p, [a], 1.55):
                    g.bond(a, g.add(p, False, 1))
                    progress = placed = True
                    break
            if not placed:
                g.cap[a] = len(g.nbr[a])  # no room for a hydrogen here: close the valence


def _grow_skeleton(rng, n_atoms):
    """Heavy-atom skeleton (rings + chains), then hydrogens on the free valences: exactly n_atoms atoms."""
    for _ in range(50):  # a growth that paints itself into a corner is restarted
        g = _Grower(rng, n_atoms)
        n_heavy = max(2, int(round(n_atoms * rng.uniform(0.42, 0.5)))) if n_atoms >= 4 else max(1, n_atoms // 2)
        g.add(np.zeros(3), True, 4 if rng.random() < 0.6 else 3)
        for _round in range(8):  # valences ran out before the budget: extend the skeleton and fill again
            _grow_heavy(g, rng, n_heavy, n_atoms)
            if _round == 0:
                # enough free valences for the hydrogens still to come?  If not, lengthen the skeleton first (placing heavy
                # atoms among hydrogens later is what clashes)
                for _ext in range(4 * n_atoms):
                    n_h = int(g.heavy[:g.n].sum())
                    free = sum(int(g.free(a)) for a in range(g.n) if g.heavy[a])
                    if free >= n_atoms - n_h or g.n >= n_atoms:
                        break
                    _grow_heavy(g, rng, n_h + 1, n_atoms)
                    if int(g.heavy[:g.n].sum()) == n_h:
                        break
            _grow_hydrogens(g, n_atoms)
            if g.n >= n_atoms:
                return g
            n_heavy = int(g.heavy[:g.n].sum()) + max(1, (n_atoms - g.n) // 3)
    raise RuntimeError(f"could not grow a {n_atoms}-atom molecule")


def _topological_distances(n, bonds):
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import shortest_path

    b = np.asarray(bonds, dtype=np.int64).reshape(-1, 2)
    adj = coo_matrix((np.ones(len(b)), (b[:, 0], b[:, 1])), shape=(n, n)).tocsr()
    d = shortest_path(adj, directed=False, unweighted=True)
    return np.where(np.isfinite(d), d, 1.0e6).astype(np.int64)  # atoms of different fragments (salts): "far apart"


def _d14(r12, r23, r34, a123, a234, phi):
    """1-4 distance for bond lengths, angles (rad) and dihedral phi (atom 2 at the origin, atom 3 on +x)."""
    import math

    x1, y1 = r12 * math.cos(a123), r12 * math.sin(a123)
    s = r34 * math.sin(a234)
    x4, y4, z4 = r23 - r34 * math.cos(a234), s * math.cos(phi), s * math.sin(phi)
    return math.sqrt((x1 - x4) ** 2 + (y1 - y4) ** 2 + z4 * z4)


def druglike_molecule(rng, n_atoms: int, with_etk: bool = True, with_mmff: bool = True):
    """One synthetic drug-like molecule.  Returns a dict: ``embed`` (FlatMolecule fields), ``mmff`` (the 7 MMFF term groups),
    ``ref`` (the hidden geometry, (n, 3)), ``bounds`` (pairs, lb, ub), ``bonds``, ``heavy``."""
    g = _grow_skeleton(rng, n_atoms)
    n, ref, nbr = n_atoms, g.pos.copy(), g.nbr
    bonds = np.array(sorted(set(g.bonds)), dtype=np.int64).reshape(-1, 2)
    topo = _topological_distances(n, bonds) if len(bonds) else np.zeros((n, n), dtype=np.int64)
    iu = np.triu_indices(n, 1)
    pairs = np.stack(iu, 1).astype(np.int64)
    dist = np.linalg.norm(ref[:, None, :] - ref[None, :, :], axis=2)
    same_ring = (g.ring[:, None] >= 0) & (g.ring[:, None] == g.ring[None, :])
    angles = np.array([(i, j, k) for j in range(n) for x, i in enumerate(nbr[j]) for k in nbr[j][x + 1:]], dtype=np.int64).reshape(-1, 3)
    torsions = np.array([(i, j, k, l) for j, k in bonds for i in nbr[j] if i != k for l in nbr[k] if l != j and l != i],
                        dtype=np.int64).reshape(-1, 4)

    def angle_of(t):
        a, b = ref[t[:, 0]] - ref[t[:, 1]], ref[t[:, 2]] - ref[t[:, 1]]
        return np.arccos(np.clip((a * b).sum(1) / np.linalg.norm(a, axis=1) / np.linalg.norm(b, axis=1), -1, 1))

    ang = angle_of(angles) if len(angles) else np.zeros(0)

    # ---- distance bounds (the shape of RDKit's setTopolBounds + triangle smoothing) ------------------------------
    lb = np.zeros((n, n))
    ub = np.full((n, n), 1000.0)
    np.fill_diagonal(ub, 0.0)
    hv = g.heavy[:n]
    floor = np.where(hv[:, None] & hv[None, :], 3.0, np.where(hv[:, None] | hv[None, :], 2.5, 2.0))
    floor = np.where(topo == 4, floor * 0.8, floor)
    lb[:] = np.minimum(floor, dist - 0.1)
    for i, j in bonds:
        lb[i, j] = lb[j, i] = dist[i, j] - 0.01
        ub[i, j] = ub[j, i] = dist[i, j] + 0.01
    for (i, j, k) in angles:
        lb[i, k] = lb[k, i] = dist[i, k] - 0.04
        ub[i, k] = ub[k, i] = dist[i, k] + 0.04
    amap = {(int(i), int(j), int(k)): a for (i, j, k), a in zip(angles, ang)}
    amap.update({(k, j, i): a for (i, j, k), a in list(amap.items())})
    for (i, j, k, l) in torsions:
        if topo[i, l] != 3:
            continue
        if same_ring[j, k] or same_ring[i, l]:
            lo, hi = dist[i, l] - 0.06, dist[i, l] + 0.06
        else:
            a1, a2 = amap[(int(i), int(j), int(k))], amap[(int(j), int(k), int(l))]
            lo = _d14(dist[i, j], dist[j, k], dist[k, l], a1, a2, 0.0) - 0.06
            hi = _d14(dist[i, j], dist[j, k], dist[k, l], a1, a2, np.pi) + 0.06
        lb[i, l] = lb[l, i] = max(min(lo, dist[i, l] - 0.02), 0.5)
        ub[i, l] = ub[l, i] = max(hi, dist[i, l] + 0.02)
    rr = same_ring & (topo >= 3)
    lb[rr] = np.maximum(dist[rr] - 0.06, 0.5)
    ub[rr] = dist[rr] + 0.06
    for k in range(n):  # triangle smoothing: upper bounds (shortest paths), then lower bounds
        ub = np.minimum(ub, ub[:, k, None] + ub[None, k, :])
    for k in range(n):
        lb = np.maximum(lb, np.maximum(lb[:, k, None] - ub[None, k, :], lb[None, k, :] - ub[:, k, None]))
    lbp, ubp = lb[iu], ub[iu]
    tp = topo[iu]

    # ---- chiral centres and stereo checks --------------------------------------------------------------------------
    checks, chiral_idx, chiral_par = [], [], []
    for a in range(n):
        if hv[a] and len(nbr[a]) == 4:
            nb = nbr[a]
            checks.append((0, (a, nb[0], nb[1], nb[2], nb[3]), (0.0,)))                    # tetrahedral centre
            p4 = ref[nb[3]]
            vol = float(np.dot(ref[nb[0]] - p4, np.cross(ref[nb[1]] - p4, ref[nb[2]] - p4)))
            if abs(vol) > 6.0 and rng.random() < 0.5:                                     # a specified stereo centre
                lo, hi = (5.0, 100.0) if vol > 0 else (-100.0, -5.0)
                chiral_idx.append(nb[:4])
                chiral_par.append((lo, hi))
                checks.append((1, (0, nb[0], nb[1], nb[2], nb[3]), (lo, hi)))
                checks.append((3, (a, nb[0], nb[1], nb[2], nb[3]), ()))
                for x in range(4):
                    for y in range(x + 1, 4):
                        checks.append((2, (nb[x], nb[y]), (lb[nb[x], nb[y]], ub[nb[x], nb[y]])))
        if hv[a] and g.cap[a] == 3 and len(nbr[a]) == 3:
            checks.append((5, (nbr[a][0], a, nbr[a][1]), ()))                              # not linear
    dg = [(pairs, np.stack([lbp**2, ubp**2, np.ones(len(pairs))], 1)),
          (np.array(chiral_idx, dtype=np.int64).reshape(-1, 4), np.array(chiral_par, dtype=np.float64).reshape(-1, 2)),
          (np.arange(n, dtype=np.int64).reshape(-1, 1), np.zeros((n, 0)))]

    # ---- ETK terms -------------------------------------------------------------------------------------------------
    etk, n_imp = None, 0
    if with_etk:
        t_idx, t_par = [], []
        seen = set()
        for (i, j, k, l) in torsions:
            if (j, k) in seen or same_ring[j, k] or not (hv[i] and hv[l]):
                continue
            seen.add((j, k))
            cj, ck = g.cap[j], g.cap[k]
            fc = np.zeros(6)
            sg = np.ones(6)
            if cj == 4 and ck == 4:
                fc[2] = rng.uniform(2.0, 6.0)
            elif cj == 3 and ck == 3:
                fc[1], sg[1] = rng.uniform(3.0, 8.0), -1.0
            else:
                fc[int(rng.integers(0, 6))] = rng.uniform(0.5, 3.0)
                fc[2] += rng.uniform(0.0, 2.0)
                sg = rng.choice([-1.0, 1.0], size=6)
            t_idx.append((i, j, k, l))
            t_par.append(np.concatenate([fc, sg]))
        imp_idx, imp_par = [], []
        for a in range(n):
            if hv[a] and g.cap[a] == 3 and len(nbr[a]) == 3:
                n_imp += 1
                x, y, z = nbr[a]
                for (p, q, r) in ((x, y, z), (x, z, y), (y, z, x)):
                    imp_idx.append((p, a, q, r))
                    imp_par.append((1.0, -1.0, 0.0, 10.0))

        def fb(mask, tol, k, pinned):
            sel = pairs[mask]
            d = dist[iu][mask]
            if tol is None:
                lo, hi = lbp[mask], ubp[mask]
            else:
                lo, hi = d - tol, d + tol
            return sel, np.stack([lo, hi, np.full(len(sel), k), np.full(len(sel), pinned)], 1)

        lin = ang > np.deg2rad(175.0) if len(ang) else np.zeros(0, dtype=bool)
        etk = [(np.array(t_idx, dtype=np.int64).reshape(-1, 4), np.array(t_par).reshape(-1, 12)),
               (np.array(imp_idx, dtype=np.int64).reshape(-1, 4), np.array(imp_par).reshape(-1, 4)),
               fb(tp == 1, 0.01, 100.0, 0.0), fb(tp == 2, 0.01, 100.0, 0.0),
               (angles[lin], np.tile([179.0, 180.0], (int(lin.sum()), 1))),
               fb(tp >= 3, None, 10.0, 0.0)]
    embed = dict(n_atoms=n, dg=dg, etk=etk, checks=_stereo_checks(checks), num_impropers=n_imp)
    out = dict(embed=embed, ref=ref, bounds=(pairs, lbp, ubp), bonds=bonds, heavy=hv.copy())

    # ---- MMFF94-shaped tables: rest values from the geometry, force constants from the ranges of the real tables --
    if with_mmff:
        b_h = ~(hv[bonds[:, 0]] & hv[bonds[:, 1]]) if len(bonds) else np.zeros(0, dtype=bool)
        bond_par = np.stack([dist[bonds[:, 0], bonds[:, 1]] * (1.0 + rng.normal(scale=0.004, size=len(bonds))),
                             np.where(b_h, rng.uniform(4.5, 5.5, len(bonds)), rng.uniform(4.0, 9.0, len(bonds)))], 1) \
            if len(bonds) else np.zeros((0, 2))
        th0 = np.degrees(ang) + rng.normal(scale=1.5, size=len(ang))
        ang_par = np.stack([th0, rng.uniform(0.4, 1.1, len(ang)), np.zeros(len(ang))], 1) if len(ang) else np.zeros((0, 3))
        sb_par = np.stack([th0, dist[angles[:, 0], angles[:, 1]], dist[angles[:, 2], angles[:, 1]],
                           rng.uniform(0.0, 0.5, len(ang)), rng.uniform(0.0, 0.5, len(ang))], 1) if len(ang) else np.zeros((0, 5))
        oop_idx = [(nbr[a][0], a, nbr[a][1], nbr[a][2]) for a in range(n) if hv[a] and g.cap[a] == 3 and len(nbr[a]) == 3]
        oop_idx = np.array([perm for (x, a, y, z) in oop_idx for perm in ((x, a, y, z), (x, a, z, y), (y, a, z, x))],
                           dtype=np.int64).reshape(-1, 4)
        oop_par = rng.uniform(0.01, 0.15, size=(len(oop_idx), 1))
        ring_bond = same_ring[torsions[:, 1], torsions[:, 2]] if len(torsions) else np.zeros(0, dtype=bool)
        tor_par = np.stack([rng.normal(scale=0.3, size=len(torsions)),
                            np.where(ring_bond, rng.uniform(3.0, 7.0, len(torsions)), rng.normal(scale=0.5, size=len(torsions))),
                            rng.uniform(0.0, 0.6, len(torsions))], 1) if len(torsions) else np.zeros((0, 3))
        far = pairs[tp >= 3]
        rstar = np.where(hv, rng.uniform(3.4, 4.0, n), rng.uniform(2.6, 3.0, n))
        epsv = np.where(hv, rng.uniform(0.04, 0.12, n), rng.uniform(0.015, 0.03, n))
        vdw_par = np.stack([0.5 * (rstar[far[:, 0]] + rstar[far[:, 1]]), np.sqrt(epsv[far[:, 0]] * epsv[far[:, 1]])], 1) \
            if len(far) else np.zeros((0, 2))
        q = rng.normal(scale=0.08, size=n)
        q -= q.mean()
        ele_par = np.stack([q[far[:, 0]] * q[far[:, 1]], np.ones(len(far)), (topo[far[:, 0], far[:, 1]] == 3).astype(float)], 1) \
            if len(far) else np.zeros((0, 3))
        out["mmff"] = [(bonds, bond_par), (angles, ang_par), (angles, sb_par), (oop_idx, oop_par), (torsions, tor_par),
                       (far, vdw_par), (far, ele_par)]
    return out


def druglike_sizes(rng, n_mols: int, mean_atoms: float = 48.0, sd: float = 12.0, lo: int = 12, hi: int = 96) -> np.ndarray:
    """Atom counts (hydrogens included) of the synthetic benchmark set: clipped Normal(48, 12) in [12, 96]."""
    return np.clip(rng.normal(mean_atoms, sd, size=n_mols).round().astype(int), lo, hi)


def _druglike_worker(args):
    seed, sizes = args
    rng = np.random.default_rng(seed)
    return [druglike_molecule(rng, int(s)) for s in sizes]


def druglike_library(n_mols: int, seed: int = 20260926, mean_atoms: float = 48.0, processes: int | None = None):
    """`n_mols` drug-like molecules, generated in parallel worker processes (the generator is per-atom Python: about 10 ms
    per molecule on one core).  Deterministic for (n_mols, seed): molecules are made in fixed chunks of 64 with per-chunk
    seeds, whatever the process count."""
    import os

    sizes = druglike_sizes(np.random.default_rng(seed), n_mols, mean_atoms)
    chunks = [(seed + 1 + c, sizes[lo:lo + 64]) for c, lo in enumerate(range(0, n_mols, 64))]
    procs = processes if processes is not None else min(len(chunks), max(1, (os.cpu_count() or 1) // 2), 64)
    if procs <= 1 or len(chunks) == 1:
        parts = [_druglike_worker(c) for c in chunks]
    else:
        import multiprocessing as mp

        with mp.get_context("fork").Pool(procs) as pool:
            parts = pool.map(_druglike_worker, chunks)
    return [m for p in parts for m in p]


# ---- real molecular graphs, synthetic parameters: BASELINE configs[2] on the reference's own molecules ----------------
#
# benchmarks/etkdg_bench.py:154-161 embeds the molecules of benchmarks/data/chembl_10k.smi after AddHs
# (benchmarks/bench_utils/molprep.py:21-55).  The library's SMILES ingestion yields their graphs (atoms, bonds, ring
# membership, hydrogen counts from RDKit's valence model); what is missing without RDKit is the parameter data (UFF bond
# lengths for the bounds matrix, the experimental-torsion patterns, MMFF94 typing).  graph_molecule() therefore derives every
# table from the REAL topology with generic parameters, the way RDKit's setTopolBounds does it from the graph alone: 1-2
# bounds from covalent radii and bond orders, 1-3 from hybridisation / ring-size angles, 1-4 from the cis / trans extremes of
# the torsion (ring and planar cases pinned), van der Waals floors beyond, triangle smoothing; chiral-centre candidates,
# impropers at planar centres, torsion preferences by hybridisation; MMFF94-shaped terms with rest values = the same ideal
# lengths and angles.  No hidden 3-D geometry is involved: whether an embedding succeeds is decided by the bounds, as in RDKit.

_COVALENT = {1: 0.31, 5: 0.84, 6: 0.76, 7: 0.71, 8: 0.66, 9: 0.57, 14: 1.11, 15: 1.07, 16: 1.05, 17: 1.02, 33: 1.19, 34: 1.20,
             35: 1.20, 53: 1.39}
_ORDER_SHORTENING = {1: 0.0, 2: 0.18, 3: 0.32, 4: 0.40, 12: 0.12}


def _graph_rings(n, nbr, ring_bonds, max_size=8):
    """The smallest ring through every ring bond (BFS with the bond removed), as sorted tuples; rings beyond max_size are
    treated as chains by the geometry rules."""
    rings = set()
    for a, b in ring_bonds:
        prev = {a: -1}
        frontier = [a]
        found = False
        for _depth in range(max_size - 1):
            nxt = []
            for u in frontier:
                for v in nbr[u]:
                    if (u == a and v == b) or v in prev:
                        continue
                    prev[v] = u
                    if v == b:
                        found = True
                        break
                    nxt.append(v)
                if found:
                    break
            if found:
                break
            frontier = nxt
        if found:
            path, u = [], b
            while u != -1:
                path.append(u)
                u = prev[u]
            rings.add(tuple(sorted(path)))
    return [set(r) for r in rings]


def graph_molecule(atoms, bonds, rng, with_etk: bool = True, with_mmff: bool = True, max_atoms: int | None = None):
    """Flattened ETKDG / MMFF tables of one molecule GRAPH as the SMILES ingestion returns it (``SmilesSet.graph``: atoms (n, 6)
    [Z, charge, isotope, total Hs, aromatic, in ring], bonds (m, 4) [begin, end, RDKit bond type, in ring]) with explicit
    hydrogens added.  Returns the dict of ``druglike_molecule`` without ``ref`` (there is no hidden geometry), or None when the
    molecule has more than ``max_atoms`` atoms."""
    atoms = np.asarray(atoms, dtype=np.int64).reshape(-1, 6)
    bonds = np.asarray(bonds, dtype=np.int64).reshape(-1, 4)
    n_heavy = len(atoms)
    n = n_heavy + int(atoms[:, 3].sum())
    if n < 2 or (max_atoms is not None and n > max_atoms):
        return None
    z = np.ones(n, dtype=np.int64)
    z[:n_heavy] = atoms[:, 0]
    nbr = [[] for _ in range(n)]
    order = {}
    blist = []
    for a, b, t, _r in bonds:
        a, b = int(a), int(b)
        nbr[a].append(b)
        nbr[b].append(a)
        order[(a, b)] = order[(b, a)] = int(t)
        blist.append((min(a, b), max(a, b)))
    h = n_heavy
    for a in range(n_heavy):
        for _ in range(int(atoms[a, 3])):
            nbr[a].append(h)
            nbr[h].append(a)
            order[(a, h)] = order[(h, a)] = 1
            blist.append((a, h))
            h += 1
    hv = z > 1
    aromatic = np.zeros(n, dtype=bool)
    aromatic[:n_heavy] = atoms[:, 4] != 0
    # hybridisation: 1 = sp, 2 = sp2, 3 = sp3
    hyb = np.full(n, 3, dtype=np.int64)
    for a in range(n_heavy):
        orders = [order[(a, b)] for b in nbr[a]]
        n_double = sum(1 for t in orders if t == 2)
        if any(t == 3 for t in orders) or n_double >= 2:
            hyb[a] = 1
        elif aromatic[a] or n_double == 1:
            hyb[a] = 2
    for a in range(n_heavy):  # nitrogen (and oxygen in rings) next to a planar atom is planar itself (amides, anilines, pyrrole-type)
        if hyb[a] == 3 and z[a] == 7 and len(nbr[a]) == 3 and any(hyb[b] <= 2 and z[b] > 1 for b in nbr[a]):
            hyb[a] = 2
    bonds_all = np.array(sorted(set(blist)), dtype=np.int64).reshape(-1, 2)
    ring_bonds = [(int(a), int(b)) for a, b, _t, r in bonds if r]
    rings = _graph_rings(n, nbr, ring_bonds)
    atom_rings = [[k for k, r in enumerate(rings) if a in r] for a in range(n)]

    def common_ring(*ats):
        best = None
        for k in atom_rings[ats[0]]:
            if all(a in rings[k] for a in ats[1:]) and (best is None or len(rings[k]) < len(rings[best])):
                best = k
        return best

    def bond_length(a, b):
        r = _COVALENT.get(int(z[a]), 1.3) + _COVALENT.get(int(z[b]), 1.3) - _ORDER_SHORTENING.get(order[(a, b)], 0.0)
        if order[(a, b)] == 1 and hyb[a] <= 2 and hyb[b] <= 2 and hv[a] and hv[b]:
            r -= 0.04  # conjugated single bond
        return r

    def ring_angle(size, planar):
        return {3: 60.0, 4: 90.0, 5: 108.0 if planar else 105.0, 6: 120.0 if planar else 111.0, 7: 124.0 if planar else 114.0,
                8: 126.0 if planar else 115.0}[size]

    def angle_at(i, j, k):
        r = common_ring(i, j, k)
        if r is not None:
            return ring_angle(len(rings[r]), hyb[j] <= 2)
        if hyb[j] == 1:
            return 180.0
        if atom_rings[j]:
            size = min(len(rings[q]) for q in atom_rings[j])
            inner = ring_angle(size, hyb[j] <= 2)
            if hyb[j] == 2:
                return (360.0 - inner) / 2.0
            return {3: 117.0, 4: 113.0}.get(size, 109.5)
        if hyb[j] == 2:
            return 120.0
        return 107.0 if z[j] == 7 else (105.0 if z[j] == 8 else (99.0 if z[j] == 16 and len(nbr[j]) == 2 else 109.5))

    dist12 = {}
    for a, b in bonds_all:
        dist12[(int(a), int(b))] = dist12[(int(b), int(a))] = bond_length(int(a), int(b))
    angles = np.array([(i, j, k) for j in range(n) for x, i in enumerate(nbr[j]) for k in nbr[j][x + 1:]], dtype=np.int64).reshape(-1, 3)
    ang = np.deg2rad(np.array([angle_at(int(i), int(j), int(k)) for i, j, k in angles])) if len(angles) else np.zeros(0)
    amap = {}
    for (i, j, k), a in zip(angles, ang):
        amap[(int(i), int(j), int(k))] = amap[(int(k), int(j), int(i))] = float(a)
    torsions = np.array([(i, j, k, l) for j, k in bonds_all for i in nbr[j] if i != k for l in nbr[k] if l != j and l != i],
                        dtype=np.int64).reshape(-1, 4)
    topo = _topological_distances(n, bonds_all)
    iu = np.triu_indices(n, 1)
    pairs = np.stack(iu, 1).astype(np.int64)

    # ---- bounds matrix --------------------------------------------------------------------------------------------
    lb = np.zeros((n, n))
    ub = np.full((n, n), 1000.0)
    np.fill_diagonal(ub, 0.0)
    floor = np.where(hv[:, None] & hv[None, :], 3.0, np.where(hv[:, None] | hv[None, :], 2.5, 2.0))
    floor = np.where(topo == 4, floor * 0.8, floor)
    lb[:] = floor
    np.fill_diagonal(lb, 0.0)
    for (a, b), r in dist12.items():
        lb[a, b], ub[a, b] = r - 0.01, r + 0.01
    d13 = {}
    for (i, j, k), a in zip(angles, ang):
        i, j, k = int(i), int(j), int(k)
        if topo[i, k] != 2:
            continue  # three-membered ring: the pair is bonded
        r1, r2 = dist12[(i, j)], dist12[(j, k)]
        d = float(np.sqrt(r1 * r1 + r2 * r2 - 2.0 * r1 * r2 * np.cos(a)))
        if (i, k) in d13:  # two paths (four-membered ring): keep the mean, widen
            d = 0.5 * (d + d13[(i, k)])
        d13[(i, k)] = d13[(k, i)] = d
        lb[i, k] = lb[k, i] = d - 0.04
        ub[i, k] = ub[k, i] = d + 0.04
    seen14 = {}
    for (i, j, k, l) in torsions:
        i, j, k, l = int(i), int(j), int(k), int(l)
        if topo[i, l] != 3:
            continue
        a1, a2 = amap[(i, j, k)], amap[(j, k, l)]
        r12, r23, r34 = dist12[(i, j)], dist12[(j, k)], dist12[(k, l)]
        cis, trans = _d14(r12, r23, r34, a1, a2, 0.0), _d14(r12, r23, r34, a1, a2, np.pi)
        rjk = common_ring(j, k)
        planar_bond = hyb[j] <= 2 and hyb[k] <= 2 and (order[(j, k)] in (2, 12) or (rjk is not None and aromatic[j] and aromatic[k]))
        if rjk is not None and i in rings[rjk] and l in rings[rjk]:
            size = len(rings[rjk])
            phi = 0.0 if (planar_bond or size <= 4) else np.deg2rad({5: 42.0, 6: 66.0, 7: 90.0, 8: 110.0}[size])
            lo, hi = cis - 0.06, _d14(r12, r23, r34, a1, a2, phi) + 0.06
        elif planar_bond and rjk is not None:
            # substituents on a planar ring bond: one in the ring, one outside = trans; both outside = cis
            inside = (i in rings[rjk]) + (l in rings[rjk])
            lo, hi = (trans - 0.06, trans + 0.06) if inside == 1 else (cis - 0.06, cis + 0.06)
        else:
            lo, hi = cis - 0.06, trans + 0.06
        key = (min(i, l), max(i, l))
        if key in seen14:  # several paths between the same pair: the union of their windows
            lo, hi = min(lo, seen14[key][0]), max(hi, seen14[key][1])
        seen14[key] = (lo, hi)
    for (i, l), (lo, hi) in seen14.items():
        lb[i, l] = lb[l, i] = max(lo, 0.5)
        ub[i, l] = ub[l, i] = hi
    for k in range(n):  # triangle smoothing: upper bounds (shortest paths), then lower bounds
        ub = np.minimum(ub, ub[:, k, None] + ub[None, k, :])
    for k in range(n):
        lb = np.maximum(lb, np.maximum(lb[:, k, None] - ub[None, k, :], lb[None, k, :] - ub[:, k, None]))
    lb = np.minimum(lb, 0.99 * ub)  # generic parameters can contradict each other in strained cages: never an empty window
    lbp, ubp, tp = lb[iu], ub[iu], topo[iu]

    # ---- chiral-centre candidates and stereo checks ----------------------------------------------------------------
    checks, chiral_idx, chiral_par = [], [], []
    n_imp = 0
    for a in range(n_heavy):
        deg = len(nbr[a])
        if deg == 4 and hyb[a] == 3:
            nb = nbr[a]
            # RDKit (findChiralSets) tests the tetrahedral shape only of C / N centres shared by two or more rings, none of them
            # a three-membered one; the flag marks fused small rings (the check then accepts a quarter of the volume)
            if z[a] in (6, 7) and len(atom_rings[a]) >= 2 and min(len(rings[q]) for q in atom_rings[a]) > 3:
                checks.append((0, (a, nb[0], nb[1], nb[2], nb[3]), (1.0 if min(len(rings[q]) for q in atom_rings[a]) <= 4 else 0.0,)))
            heavy_nb = sum(1 for b in nb if hv[b])
            if heavy_nb >= 3 and len(atom_rings[a]) <= 1 and z[a] in (6, 7, 14, 15, 16) and rng.random() < 0.35:
                lo, hi = (5.0, 100.0) if rng.random() < 0.5 else (-100.0, -5.0)
                chiral_idx.append(nb[:4])
                chiral_par.append((lo, hi))
                checks.append((1, (0, nb[0], nb[1], nb[2], nb[3]), (lo, hi)))
                checks.append((3, (a, nb[0], nb[1], nb[2], nb[3]), ()))
                for x in range(4):
                    for y in range(x + 1, 4):
                        checks.append((2, (nb[x], nb[y]), (lb[nb[x], nb[y]], ub[nb[x], nb[y]])))
        if deg == 3 and hyb[a] == 2:
            checks.append((5, (nbr[a][0], a, nbr[a][1]), ()))
    dg = [(pairs, np.stack([lbp**2, ubp**2, np.ones(len(pairs))], 1)),
          (np.array(chiral_idx, dtype=np.int64).reshape(-1, 4), np.array(chiral_par, dtype=np.float64).reshape(-1, 2)),
          (np.arange(n, dtype=np.int64).reshape(-1, 1), np.zeros((n, 0)))]

    etk = None
    if with_etk:
        t_idx, t_par, seen = [], [], set()
        for (i, j, k, l) in torsions:
            i, j, k, l = int(i), int(j), int(k), int(l)
            if (j, k) in seen or common_ring(j, k) is not None or not (hv[i] and hv[l]) or hyb[j] == 1 or hyb[k] == 1:
                continue
            seen.add((j, k))
            fc, sg = np.zeros(6), np.ones(6)
            if hyb[j] == 3 and hyb[k] == 3:
                fc[2] = rng.uniform(2.0, 6.0)
            elif hyb[j] == 2 and hyb[k] == 2:
                fc[1], sg[1] = rng.uniform(3.0, 8.0), -1.0
            else:
                fc[int(rng.integers(0, 6))] = rng.uniform(0.5, 3.0)
                fc[2] += rng.uniform(0.0, 2.0)
                sg = rng.choice([-1.0, 1.0], size=6)
            t_idx.append((i, j, k, l))
            t_par.append(np.concatenate([fc, sg]))
        imp_idx, imp_par = [], []
        for a in range(n_heavy):
            if hyb[a] == 2 and len(nbr[a]) == 3:
                n_imp += 1
                x, y, w = nbr[a]
                for (p, q, r) in ((x, y, w), (x, w, y), (y, w, x)):
                    imp_idx.append((p, a, q, r))
                    imp_par.append((1.0, -1.0, 0.0, 10.0))
        centre = 0.5 * (lbp + ubp)

        def fb(mask, tol, k):
            sel = pairs[mask]
            if tol is None:
                lo, hi = lbp[mask], ubp[mask]
            else:
                lo, hi = centre[mask] - tol, centre[mask] + tol
            return sel, np.stack([lo, hi, np.full(len(sel), k), np.zeros(len(sel))], 1)

        lin = ang > np.deg2rad(175.0) if len(ang) else np.zeros(0, dtype=bool)
        etk = [(np.array(t_idx, dtype=np.int64).reshape(-1, 4), np.array(t_par).reshape(-1, 12)),
               (np.array(imp_idx, dtype=np.int64).reshape(-1, 4), np.array(imp_par).reshape(-1, 4)),
               fb(tp == 1, 0.01, 100.0), fb(tp == 2, 0.01, 100.0),
               (angles[lin], np.tile([179.0, 180.0], (int(lin.sum()), 1))),
               fb(tp >= 3, None, 10.0)]
    embed = dict(n_atoms=n, dg=dg, etk=etk, checks=_stereo_checks(checks), num_impropers=n_imp)
    out = dict(embed=embed, bounds=(pairs, lbp, ubp), bonds=bonds_all, heavy=hv.copy(), elements=z.copy())

    if with_mmff:
        nb_ = len(bonds_all)
        r0 = np.array([dist12[(int(a), int(b))] for a, b in bonds_all]) if nb_ else np.zeros(0)
        b_h = ~(hv[bonds_all[:, 0]] & hv[bonds_all[:, 1]]) if nb_ else np.zeros(0, dtype=bool)
        bond_par = np.stack([r0, np.where(b_h, rng.uniform(4.5, 5.5, nb_), rng.uniform(4.0, 9.0, nb_))], 1) if nb_ else np.zeros((0, 2))
        th0 = np.degrees(ang)
        linear = (th0 > 175.0).astype(float)
        ang_par = np.stack([th0, rng.uniform(0.4, 1.1, len(ang)), linear], 1) if len(ang) else np.zeros((0, 3))
        sb_par = np.stack([th0, [dist12[(int(i), int(j))] for i, j, _k in angles], [dist12[(int(k), int(j))] for _i, j, k in angles],
                           np.where(linear > 0, 0.0, rng.uniform(0.0, 0.5, len(ang))), np.where(linear > 0, 0.0, rng.uniform(0.0, 0.5, len(ang)))], 1) \
            if len(ang) else np.zeros((0, 5))
        oop_c = [a for a in range(n_heavy) if hyb[a] == 2 and len(nbr[a]) == 3]
        oop_idx = np.array([perm for a in oop_c for (x, y, w) in [tuple(nbr[a])] for perm in ((x, a, y, w), (x, a, w, y), (y, a, w, x))],
                           dtype=np.int64).reshape(-1, 4)
        oop_par = rng.uniform(0.01, 0.15, size=(len(oop_idx), 1))
        if len(torsions):
            lin_t = np.array([hyb[int(j)] == 1 or hyb[int(k)] == 1 for _i, j, k, _l in torsions])
            tors = torsions[~lin_t]  # MMFF has no torsions about linear centres
        else:
            tors = torsions
        planar_t = np.array([hyb[int(j)] <= 2 and hyb[int(k)] <= 2 and order[(int(j), int(k))] in (2, 12) for _i, j, k, _l in tors]) \
            if len(tors) else np.zeros(0, dtype=bool)
        tor_par = np.stack([rng.normal(scale=0.3, size=len(tors)),
                            np.where(planar_t, rng.uniform(3.0, 7.0, len(tors)), rng.normal(scale=0.5, size=len(tors))),
                            rng.uniform(0.0, 0.6, len(tors))], 1) if len(tors) else np.zeros((0, 3))
        far = pairs[tp >= 3]
        rstar = np.where(hv, rng.uniform(3.4, 4.0, n), rng.uniform(2.6, 3.0, n))
        epsv = np.where(hv, rng.uniform(0.04, 0.12, n), rng.uniform(0.015, 0.03, n))
        vdw_par = np.stack([0.5 * (rstar[far[:, 0]] + rstar[far[:, 1]]), np.sqrt(epsv[far[:, 0]] * epsv[far[:, 1]])], 1) \
            if len(far) else np.zeros((0, 2))
        q = rng.normal(scale=0.08, size=n)
        q -= q.mean()
        ele_par = np.stack([q[far[:, 0]] * q[far[:, 1]], np.ones(len(far)), (topo[far[:, 0], far[:, 1]] == 3).astype(float)], 1) \
            if len(far) else np.zeros((0, 3))
        out["mmff"] = [(bonds_all, bond_par), (angles, ang_par), (angles, sb_par), (oop_idx, oop_par), (tors, tor_par),
                       (far, vdw_par), (far, ele_par)]
    return out


def _graph_worker(args):
    seed, graphs, max_atoms = args
    rng = np.random.default_rng(seed)
    return [graph_molecule(a, b, rng, max_atoms=max_atoms) for a, b in graphs]


def graph_library(graphs, seed: int = 20260927, max_atoms: int | None = None, processes: int | None = None):
    """graph_molecule() for a list of (atoms, bonds) graphs, in parallel worker processes; deterministic for (graphs, seed)
    whatever the process count (fixed chunks of 32 with per-chunk seeds).  Molecules beyond ``max_atoms`` come back as None."""
    import os

    chunks = [(seed + 1 + c, graphs[lo:lo + 32], max_atoms) for c, lo in enumerate(range(0, len(graphs), 32))]
    procs = processes if processes is not None else min(len(chunks), max(1, (os.cpu_count() or 1) // 2), 64)
    if procs <= 1 or len(chunks) <= 1:
        parts = [_graph_worker(c) for c in chunks]
    else:
        import multiprocessing as mp

        with mp.get_context("fork").Pool(procs) as pool:
            parts = pool.map(_graph_worker, chunks)
    return [m for p in parts for m in p]


def smiles_file_library(path, n_mols: int | None = None, seed: int = 20260927, max_atoms: int | None = None,
                        processes: int | None = None):
    """The molecules of a .smi file (e.g. tests/golden/chembl_10k.smi = the reference's benchmarks/data/chembl_10k.smi) through
    the library's own ingestion -> graph_molecule(): real topologies with explicit hydrogens, synthetic parameters.  Returns
    (library without the molecules that were refused or exceed max_atoms, indices of the kept molecules)."""
    from nvmolkit_amd.fingerprints import SmilesSet

    s = SmilesSet.from_file(str(path))
    ids = [i for i in range(len(s.status)) if s.status[i] == 0][:n_mols]
    graphs = [s.graph(i) for i in ids]
    lib = graph_library(graphs, seed=seed, max_atoms=max_atoms, processes=processes)
    keep = [k for k, m in enumerate(lib) if m is not None]
    return [lib[k] for k in keep], [ids[k] for k in keep]
