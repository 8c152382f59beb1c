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
