"""CPU restatement (numpy) of the ETKDG stereochemistry checks — TEST INFRASTRUCTURE ONLY.

Follows the reference's per-term kernels in src/etkdg_stage_stereochem_checks.cu (themselves ports of RDKit's embedder
checks): `_sameSide` :25-50, tetrahedral volume + centre-in-volume :52-182 (MIN_TETRAHEDRAL_CHIRAL_VOL 0.50 :22,
tolerances 0.3 / 0.1 from etkdg_stage_stereochem_checks.h:69,122), first chiral check :229-259, chiral distance matrix
:261-301, double-bond stereo :303-377, double-bond geometry :379-442.  Term layout = the C ABI's (include/nvmolkit_amd.h):
kind, 5 local atom indices, 2 doubles.  `fails(...)` returns True when the reference would set failedThisStage.

Pinned by hand-computed cases in tests/test_oracle_stereo.py (regular tetrahedron, planar centre, centre outside its
neighbours' volume, cis / trans / perpendicular dihedrals, collinear double bond, chiral-volume and distance windows); the
reference's own tests compare against RDKit at run time and hold no vectors for this row.
"""

from __future__ import annotations

import math

import numpy as np

TETRAHEDRAL, CHIRAL_VOLUME, CHIRAL_DISTANCE, CHIRAL_CENTER_VOLUME, DOUBLE_BOND_STEREO, DOUBLE_BOND_GEOMETRY = range(6)
MIN_TETRAHEDRAL_CHIRAL_VOL = 0.50


def _same_side(tol, v1, v2, v3, v4, p0) -> bool:
    n = np.cross(v2 - v1, v3 - v1)
    d1 = float(np.dot(n, v4 - v1))
    d2 = float(np.dot(n, p0 - v1))
    if abs(d1) < tol or abs(d2) < tol:
        return False
    return not ((d1 < 0.0) ^ (d2 < 0.0))


def _unit(v):
    # no guard, as in the reference (normalizeVector, src/forcefields/kernel_utils.cuh:140-145): a zero vector becomes NaN and
    # every comparison with it is false — that is how a three-coordinate centre (idx4 == idx0) passes the volume test
    with np.errstate(invalid="ignore", divide="ignore"):
        return v / np.float64(math.sqrt(float(np.dot(v, v))))


def _tetrahedral_ok(p, ix, volume_test: bool, fused_small_rings: bool, tol: float) -> bool:
    p0, p1, p2, p3, p4 = (p[i] for i in ix)
    if volume_test:
        d1, d2, d3, d4 = _unit(p0 - p1), _unit(p0 - p2), _unit(p0 - p3), _unit(p0 - p4)
        lim = (0.25 if fused_small_rings else 1.0) * MIN_TETRAHEDRAL_CHIRAL_VOL
        for a, b, c in ((d1, d2, d3), (d1, d2, d4), (d1, d3, d4), (d2, d3, d4)):
            if abs(float(np.dot(np.cross(a, b), c))) < lim:
                return False
    if ix[0] == ix[4]:  # three-coordinate centre: no centre-in-volume test
        return True
    return (_same_side(tol, p1, p2, p3, p4, p0) and _same_side(tol, p2, p3, p4, p1, p0) and
            _same_side(tol, p3, p4, p1, p2, p0) and _same_side(tol, p4, p1, p2, p3, p0))


def fails(kind: int, pos: np.ndarray, idx, par) -> bool:
    """pos: (n_atoms, >=3) coordinates of one system (only x, y, z are used); idx: 5 local indices; par: 2 doubles."""
    p = np.asarray(pos, dtype=np.float64)[:, :3]
    ix = [int(v) for v in idx]
    a, b = float(par[0]), float(par[1])
    if kind == TETRAHEDRAL:
        return not _tetrahedral_ok(p, ix, True, a != 0.0, 0.3)
    if kind == CHIRAL_CENTER_VOLUME:
        return not _tetrahedral_ok(p, ix, False, False, 0.1)
    if kind == CHIRAL_VOLUME:
        p4 = p[ix[4]]
        vol = float(np.dot(p[ix[1]] - p4, np.cross(p[ix[2]] - p4, p[ix[3]] - p4)))
        opp = lambda x, y: math.copysign(1.0, x) != math.copysign(1.0, y)  # noqa: E731
        return bool((a > 0 and vol < a and (vol / a < 0.8 or opp(vol, a))) or
                    (b < 0 and vol > b and (vol / b < 0.8 or opp(vol, b))))
    if kind == CHIRAL_DISTANCE:
        d = p[ix[0]] - p[ix[1]]
        dist = math.sqrt(float(np.dot(d, d)))
        return bool((dist < a and abs(dist - a) > 0.1 * b) or (dist > b and abs(dist - b) > 0.1 * b))
    if kind == DOUBLE_BOND_STEREO:
        p0, p1, p2, p3 = p[ix[0]], p[ix[1]], p[ix[2]], p[ix[3]]
        r1 = p2 - p1
        c1, c2 = np.cross(p0 - p1, r1), np.cross(p3 - p2, r1)
        den = math.sqrt(float(np.dot(c1, c1)) * float(np.dot(c2, c2)))
        dot = float(np.dot(c1, c2)) / den if den != 0.0 else float("nan")  # collinear substituent: the device divides 0 by 0
        angle = math.pi if dot <= -1.0 else (0.0 if dot >= 1.0 else (math.acos(dot) if dot == dot else float("nan")))
        return (angle - math.pi / 2) * a < 0.0
    if kind == DOUBLE_BOND_GEOMETRY:
        u, v = _unit(p[ix[1]] - p[ix[0]]), _unit(p[ix[1]] - p[ix[2]])
        return float(np.dot(u, v)) + 1.0 < 1.0e-3
    raise ValueError(f"unknown check kind {kind}")


def system_fails(kind: int, pos: np.ndarray, kinds, idx, par) -> bool:
    """True if any term of the given kind fails on this system (the stage's failedThisStage flag)."""
    return any(int(k) == kind and fails(kind, pos, idx[t], par[t]) for t, k in enumerate(kinds))
