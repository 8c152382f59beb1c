"""CPU oracle for the force-field / BFGS path — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy restatements of the term energies (vectorised over the terms of ONE system), a finite-difference
gradient, and a plain-Python RDKit-style BFGS.  Each function cites the reference lines it follows.

Pinning status: "parity unpinned" against RDKit force-field contribs (RDKit's typing / parameter tables are
needed to produce real term lists; neither the build nor the GPU image has RDKit).  Pinned without RDKit:
  * analytic gradients of the product against central finite differences of THESE energies (tolerance 1e-6
    relative), with the two RDKit conventions where the "gradient" is half the derivative (chiral volume,
    fourth dimension) applied explicitly;
  * closed-form values of single terms (zero inside bounds, known geometry) in tests/test_oracle_ff.py;
  * UFF: every angle order, torsion periodicity and the vdW cutoff in closed form; the reference's analytic inversion
    gradient restated (uff_inversion_gradient_reference) and shown to equal the finite difference of the energy with
    the C2 part negated, which is the convention system_gradient(UFF) applies;
  * BFGS: the reference's RDKit-free quartic test (tests/test_bfgs_minimizer.cu:823-1029, converge to x_p = p).
"""

from __future__ import annotations

import numpy as np

RAD2DEG = 180.0 / np.pi
DEG2RAD = np.pi / 180.0
MDYNE_A_TO_KCAL = 143.9325

# group layouts (n_idx, n_par) per force-field kind — must match include/nvmolkit_amd.h
DG, ETK, MMFF, QUARTIC, UFF = 0, 1, 2, 3, 4
LAYOUT = {
    DG: [(2, 3), (4, 2), (1, 0)],
    ETK: [(4, 12), (4, 4), (2, 4), (2, 4), (3, 2), (2, 4)],
    MMFF: [(2, 2), (3, 3), (3, 5), (4, 1), (4, 3), (2, 2), (2, 3)],
    QUARTIC: [],
    UFF: [(2, 2), (3, 6), (4, 3), (4, 4), (2, 3)],
}
DIM = {DG: 4, ETK: 3, MMFF: 3, QUARTIC: 4, UFF: 3}
# optional constraint groups appended to the MMFF / UFF groups: distance, position, angle, torsion
CONSTRAINT_LAYOUT = [(2, 3), (1, 5), (3, 3), (4, 3)]


def _xyz(pos, idx):
    return pos[idx][:, :3]


def _cos_angle(p1, p2, p3):
    r1, r2 = p1 - p2, p3 - p2
    l1, l2 = (r1 * r1).sum(1), (r2 * r2).sum(1)
    ok = (l1 > 1e-16) & (l2 > 1e-16)
    c = np.zeros(len(p1))
    c[ok] = np.clip((r1[ok] * r2[ok]).sum(1) / np.sqrt(l1[ok] * l2[ok]), -1.0, 1.0)
    return c, ok


def _cos_dihedral(p1, p2, p3, p4):
    r1, r2, r4 = p1 - p2, p3 - p2, p4 - p3
    t1, t2 = np.cross(r1, r2), np.cross(-r2, r4)
    d = (t1 * t1).sum(1) * (t2 * t2).sum(1)
    ok = d > 1e-16
    c = np.zeros(len(p1))
    c[ok] = np.clip((t1[ok] * t2[ok]).sum(1) / np.sqrt(d[ok]), -1.0, 1.0)
    return c, ok


# ---- DG (src/forcefields/dist_geom_kernels_device.cuh:37-231) ---------------------------------------

def dg_terms(pos, groups, chiral_w, fourth_w):
    """Per-group energy arrays of one system; pos is (n_atoms, 4)."""
    out = []
    idx, par = groups[0]
    d2 = ((pos[idx[:, 0]] - pos[idx[:, 1]]) ** 2).sum(1)  # all four dimensions (:43-45)
    lb2, ub2, w = par[:, 0], par[:, 1], par[:, 2]
    val = np.where(d2 > ub2, d2 / ub2 - 1.0, np.where(d2 < lb2, 2.0 * lb2 / (lb2 + d2) - 1.0, 0.0))
    out.append(w * np.maximum(val, 0.0) ** 2)
    idx, par = groups[1]
    p1, p2, p3, p4 = (_xyz(pos, idx[:, k]) for k in range(4))
    vol = ((p1 - p4) * np.cross(p2 - p4, p3 - p4)).sum(1)  # (:97-130)
    lo, hi = par[:, 0], par[:, 1]
    out.append(chiral_w * np.where(vol < lo, (vol - lo) ** 2, np.where(vol > hi, (vol - hi) ** 2, 0.0)))
    idx, _ = groups[2]
    out.append(fourth_w * pos[idx[:, 0], 3] ** 2)  # (:209-217)
    return out


# ---- ETK (dist_geom_kernels_device.cuh:237-445) -----------------------------------------------------

def _flat_bottom_dist(pos, idx, par):
    d = np.sqrt(((_xyz(pos, idx[:, 0]) - _xyz(pos, idx[:, 1])) ** 2).sum(1))
    lo, hi, k = par[:, 0], par[:, 1], par[:, 2]
    diff = np.where(d < lo, lo - d, np.where(d > hi, d - hi, 0.0))
    return 0.5 * k * diff * diff  # (:368-392)


def etk_terms(pos, groups):
    out = []
    idx, par = groups[0]
    c, _ = _cos_dihedral(*(_xyz(pos, idx[:, k]) for k in range(4)))  # degenerate -> cosPhi = 0 (:286-288)
    cheb = [c, 2 * c**2 - 1, 4 * c**3 - 3 * c, 8 * c**4 - 8 * c**2 + 1, 16 * c**5 - 20 * c**3 + 5 * c,
            32 * c**6 - 48 * c**4 + 18 * c**2 - 1]
    out.append(sum(par[:, k] * (1.0 + par[:, 6 + k] * cheb[k]) for k in range(6)))  # (:237-258)
    idx, par = groups[1]
    p1, p2, p3, p4 = (_xyz(pos, idx[:, k]) for k in range(4))
    rji, rjk, rjl = p1 - p2, p3 - p2, p4 - p2
    n = np.cross(rji, rjk)
    ln, ll = (n * n).sum(1), (rjl * rjl).sum(1)
    ok = (ln > 1e-16 * (rji * rji).sum(1) * (rjk * rjk).sum(1)) & (ll > 1e-16) & ((rji * rji).sum(1) > 1e-16) & ((rjk * rjk).sum(1) > 1e-16)
    cos_y = np.zeros(len(idx))
    cos_y[ok] = np.clip((n[ok] * rjl[ok]).sum(1) / np.sqrt(ln[ok] * ll[ok]), -1.0, 1.0)
    sin_y_sq = np.maximum(1.0 - cos_y**2, 1e-16)
    out.append(par[:, 3] * (par[:, 0] + par[:, 1] * np.sqrt(sin_y_sq) + par[:, 2] * (2.0 * sin_y_sq - 1.0)))  # (:345-366)
    out.append(_flat_bottom_dist(pos, *groups[2]))
    out.append(_flat_bottom_dist(pos, *groups[3]))
    idx, par = groups[4]
    c, ok = _cos_angle(*(_xyz(pos, idx[:, k]) for k in range(3)))
    theta = RAD2DEG * np.arccos(c)
    term = np.where(theta < par[:, 0], theta - par[:, 0], np.where(theta > par[:, 1], theta - par[:, 1], 0.0))
    out.append(np.where(ok, term * term, 0.0))  # force constant 1 (:394-445, defaultAngleForceConstant)
    out.append(_flat_bottom_dist(pos, *groups[5]))
    return out


# ---- MMFF94 (src/forcefields/mmff_kernels_device.cuh:28-660) ----------------------------------------

def mmff_terms(pos, groups):
    out = []
    idx, par = groups[0]
    r = np.sqrt(((pos[idx[:, 0]] - pos[idx[:, 1]]) ** 2).sum(1))
    dr = r - par[:, 0]
    cs = -2.0
    out.append(0.5 * MDYNE_A_TO_KCAL * par[:, 1] * dr**2 * (1.0 + cs * dr + 7.0 / 12.0 * cs * cs * dr**2))
    idx, par = groups[1]
    c, ok = _cos_angle(*(pos[idx[:, k]] for k in range(3)))
    dt = RAD2DEG * np.arccos(c) - par[:, 0]
    bend = 0.5 * MDYNE_A_TO_KCAL * DEG2RAD**2 * par[:, 1] * dt**2 * (1.0 - 0.4 * DEG2RAD * dt)
    out.append(np.where(ok, np.where(par[:, 2] != 0, MDYNE_A_TO_KCAL * par[:, 1] * (1.0 + c), bend), 0.0))
    idx, par = groups[2]
    p1, p2, p3 = (pos[idx[:, k]] for k in range(3))
    d1, d2 = np.sqrt(((p1 - p2) ** 2).sum(1)), np.sqrt(((p3 - p2) ** 2).sum(1))
    c, ok = _cos_angle(p1, p2, p3)
    dt = RAD2DEG * np.arccos(c) - par[:, 0]
    out.append(np.where(ok, 2.51210 * dt * ((d1 - par[:, 1]) * par[:, 3] + (d2 - par[:, 2]) * par[:, 4]), 0.0))
    idx, par = groups[3]
    p1, p2, p3, p4 = (pos[idx[:, k]] for k in range(4))
    n = np.cross(p1 - p2, p3 - p2)
    rjl = p4 - p2
    ln, ll = (n * n).sum(1), (rjl * rjl).sum(1)
    ok = (ln > 1e-16) & (ll > 1e-16)
    s = np.zeros(len(idx))
    s[ok] = np.clip((n[ok] * rjl[ok]).sum(1) / np.sqrt(ln[ok] * ll[ok]), -1.0, 1.0)
    chi = RAD2DEG * np.arcsin(s)
    out.append(0.5 * MDYNE_A_TO_KCAL * DEG2RAD**2 * par[:, 0] * chi**2)
    idx, par = groups[4]
    c, _ = _cos_dihedral(*(pos[idx[:, k]] for k in range(4)))
    out.append(0.5 * (par[:, 0] * (1.0 + c) + par[:, 1] * (1.0 - (2 * c**2 - 1)) + par[:, 2] * (1.0 + (4 * c**3 - 3 * c))))
    idx, par = groups[5]
    r = np.sqrt(((pos[idx[:, 0]] - pos[idx[:, 1]]) ** 2).sum(1))
    rs, eps = par[:, 0], par[:, 1]
    out.append(eps * (1.07 * rs / (r + 0.07 * rs)) ** 7 * (1.12 * rs**7 / (r**7 + 0.12 * rs**7) - 2.0))
    idx, par = groups[6]
    r = np.sqrt(((pos[idx[:, 0]] - pos[idx[:, 1]]) ** 2).sum(1)) + 0.05
    e = 332.0716 * par[:, 0] / np.where(par[:, 1] == 2, r * r, r)
    out.append(np.where(par[:, 2] != 0, 0.75 * e, e))
    return out


# ---- UFF (src/forcefields/uff_kernels_device.cuh:37-580) ---------------------------------------------

def _uff_sin_y(p1, p2, p3, p4):
    """sinY of the inversion term (uffCalculateCosY :391-424 + :437-438); degenerate geometry -> cosY = 0."""
    rji, rjk, rjl = p1 - p2, p3 - p2, p4 - p2
    li, lk, ll = (rji * rji).sum(1), (rjk * rjk).sum(1), (rjl * rjl).sum(1)
    n = np.cross(rji, rjk)
    ln = (n * n).sum(1)
    ok = (li >= 1e-16) & (lk >= 1e-16) & (ll >= 1e-16) & (ln >= 1e-16 * li * lk)
    cos_y = np.zeros(len(p1))
    cos_y[ok] = np.clip((n[ok] * rjl[ok]).sum(1) / np.sqrt(ln[ok] * ll[ok]), -1.0, 1.0)
    return np.sqrt(np.maximum(1.0 - cos_y * cos_y, 0.0))


def uff_terms(pos, groups, gradient_convention: bool = False):
    """Per-group energies.  gradient_convention=True flips the sign of the inversion C2 part: the finite difference
    of THAT function is what the reference's analytic inversion gradient computes (its dE/dW,
    uff_kernels_device.cuh:497, has the C2 part with the opposite sign of the derivative of its own energy;
    `uff_inversion_gradient_reference` below restates the formula and tests/test_oracle_ff.py checks the claim)."""
    out = []
    idx, par = groups[0]
    r = np.sqrt(((pos[idx[:, 0]] - pos[idx[:, 1]]) ** 2).sum(1))
    out.append(0.5 * par[:, 1] * (r - par[:, 0]) ** 2)                                           # :37-45
    idx, par = groups[1]
    c, ok = _cos_angle(*(pos[idx[:, k]] for k in range(3)))
    s2 = 1.0 - c * c
    theta0, k, order = par[:, 0], par[:, 1], par[:, 2].astype(np.int64)
    c2t = c * c - s2
    f = np.select([order == 1, order == 2, order == 3, order == 4],
                  [-c, c2t, c * (c * c - 3.0 * s2), c**4 - 6.0 * c * c * s2 + s2 * s2], default=0.0)
    e = np.where(order == 0, par[:, 3] + par[:, 4] * c + par[:, 5] * c2t, (1.0 - f) / np.maximum(order, 1) ** 2) * k  # :78-108
    corr = (order > 0) & (order < 5) & (c > 0.8660)                                               # :167-170
    e = e + np.where(corr, np.exp(-20.0 * (np.arccos(c) - theta0 + 0.25)), 0.0)
    out.append(np.where(ok, e, 0.0))
    idx, par = groups[2]
    c, _ = _cos_dihedral(*(pos[idx[:, k]] for k in range(4)))
    s2 = 1.0 - c * c
    order = par[:, 1].astype(np.int64)
    cn = np.select([order == 2, order == 3, order == 6],
                   [1.0 - 2.0 * s2, c * (c * c - 3.0 * s2), 1.0 + s2 * (-32.0 * s2 * s2 + 48.0 * s2 - 18.0)], default=np.nan)
    out.append(np.where(np.isnan(cn), 0.0, 0.5 * par[:, 0] * (1.0 - par[:, 2] * np.nan_to_num(cn))))  # :302-326
    idx, par = groups[3]
    sin_y = _uff_sin_y(*(pos[idx[:, k]] for k in range(4)))
    c2sign = -1.0 if gradient_convention else 1.0
    out.append(par[:, 0] * (par[:, 1] + par[:, 2] * sin_y + c2sign * par[:, 3] * (2.0 * sin_y * sin_y - 1.0)))  # :426-440
    idx, par = groups[4]
    r = np.sqrt(((pos[idx[:, 0]] - pos[idx[:, 1]]) ** 2).sum(1))
    inside = (r <= par[:, 2]) & (r > 0.0)
    q6 = (par[:, 0] / np.where(inside, r, 1.0)) ** 6
    out.append(np.where(inside, par[:, 1] * (q6 * q6 - 2.0 * q6), 0.0))                           # :527-540
    return out


def uff_inversion_gradient_reference(pos, idx, par):
    """The reference's analytic inversion gradient, restated term by term (uff_kernels_device.cuh:442-525), for ONE
    term: idx (4,), par (k, C0, C1, C2).  Returns (4, 3)."""
    p1, p2, p3, p4 = (pos[i][:3] for i in idx)
    k, _, c1, c2 = par
    rji, rjk, rjl = p1 - p2, p3 - p2, p4 - p2
    dji, djk, djl = (np.linalg.norm(v) for v in (rji, rjk, rjl))
    rji, rjk, rjl = rji / dji, rjk / djk, rjl / djl
    n = np.cross(-rji, rjk)
    n = n / np.linalg.norm(n)
    cos_y = float(np.clip(n.dot(rjl), -1.0, 1.0))
    sin_y = max(np.sqrt(1.0 - cos_y * cos_y), 1e-8)
    cos_t = float(np.clip(rji.dot(rjk), -1.0, 1.0))
    sin_t_sq = 1.0 - cos_t * cos_t
    sin_t = max(np.sqrt(sin_t_sq), 1e-8)
    de_dw = -k * (c1 * cos_y - 4.0 * c2 * cos_y * sin_y)
    t1, t2, t3 = np.cross(rjl, rjk), np.cross(rji, rjl), np.cross(rjk, rji)
    term1, term2 = sin_y * sin_t, cos_y / (sin_y * sin_t_sq)
    tg1 = (t1 / term1 - (rji - rjk * cos_t) * term2) / dji
    tg3 = (t2 / term1 - (rjk - rji * cos_t) * term2) / djk
    tg4 = (t3 / term1 - rjl * cos_y / sin_y) / djl
    return np.array([de_dw * tg1, -de_dw * (tg1 + tg3 + tg4), de_dw * tg3, de_dw * tg4])


# ---- constraints shared by MMFF and UFF (src/forcefields/mmff_kernels_device.cuh:663-1036) -------------

def _normalize_deg(a):
    a = np.fmod(a, 360.0)
    return np.where(a < -180.0, a + 360.0, np.where(a > 180.0, a - 360.0, a))


def signed_dihedral_deg(p1, p2, p3, p4):
    """computeSignedDihedral (:899-960), vectorised over terms."""
    r0, r1, r3 = p1 - p2, p3 - p2, p4 - p3
    r2 = -r1
    t0, t1 = np.cross(r0, r1), np.cross(r2, r3)
    t0 = t0 / np.maximum(np.linalg.norm(t0, axis=1), 1e-5)[:, None]
    t1 = t1 / np.maximum(np.linalg.norm(t1, axis=1), 1e-5)[:, None]
    cos_phi = np.clip((t0 * t1).sum(1), -1.0, 1.0)
    m = np.cross(t0, r1)
    ml = np.maximum(np.linalg.norm(m, axis=1), 1e-5)
    return -np.arctan2((m * t1).sum(1) / ml, cos_phi) * RAD2DEG


def dihedral_window_offset(dihedral, lo, hi):
    """computeDihedralConstraintTerm (:879-897)."""
    inside = ((dihedral > lo) & (dihedral < hi)) | ((dihedral > lo) & (lo > hi)) | ((dihedral < hi) & (lo > hi))
    to_lo, to_hi = _normalize_deg(dihedral - lo), _normalize_deg(dihedral - hi)
    target = np.where(inside, dihedral, np.where(np.abs(to_lo) < np.abs(to_hi), lo, hi))
    return _normalize_deg(dihedral - target)


def constraint_terms(pos, groups):
    """Per-group energies of the four optional constraint groups (any may be missing / empty)."""
    out = []
    groups = list(groups) + [(np.zeros((0, n), dtype=np.int64), np.zeros((0, m))) for n, m in CONSTRAINT_LAYOUT[len(groups):]]
    idx, par = groups[0]
    d = np.sqrt(((pos[idx[:, 0]] - pos[idx[:, 1]]) ** 2).sum(1))
    diff = np.where(d < par[:, 0], par[:, 0] - d, np.where(d > par[:, 1], d - par[:, 1], 0.0))
    out.append(0.5 * par[:, 2] * diff**2)                                                         # :674-690
    idx, par = groups[1]
    dist = np.sqrt(((pos[idx[:, 0]] - par[:, :3]) ** 2).sum(1))
    out.append(0.5 * par[:, 4] * np.maximum(dist - par[:, 3], 0.0) ** 2)                          # :720-733
    idx, par = groups[2]
    r1, r2 = pos[idx[:, 0]] - pos[idx[:, 1]], pos[idx[:, 2]] - pos[idx[:, 1]]
    l1, l2 = np.maximum((r1 * r1).sum(1), 1e-5), np.maximum((r2 * r2).sum(1), 1e-5)
    theta = RAD2DEG * np.arccos(np.clip((r1 * r2).sum(1) / np.sqrt(l1 * l2), -1.0, 1.0))
    term = np.where(theta < par[:, 0], theta - par[:, 0], np.where(theta > par[:, 1], theta - par[:, 1], 0.0))
    out.append(par[:, 2] * term**2)                                                               # :770-815
    idx, par = groups[3]
    phi = signed_dihedral_deg(*(pos[idx[:, k]] for k in range(4)))
    out.append(par[:, 2] * dihedral_window_offset(phi, par[:, 0], par[:, 1]) ** 2)                # :962-976
    return out


def system_energy(kind: int, pos: np.ndarray, groups, w0: float = 1.0, w1: float = 1.0, coord_start: int = 0,
                  per_group: bool = False):
    """Energy of one system.  pos: (n_atoms, DIM[kind]); groups: list of (idx (n, n_idx) int, par (n, n_par))."""
    if kind == QUARTIC:
        target = coord_start + np.arange(pos.size).reshape(pos.shape)
        diff = pos - target
        if w0 == 0.0:
            diff = diff[:, :3]
        return float((diff**4).sum())
    n_base = len(LAYOUT[kind])
    parts = {DG: lambda: dg_terms(pos, groups, w0, w1), ETK: lambda: etk_terms(pos, groups),
             MMFF: lambda: mmff_terms(pos, groups[:n_base]), UFF: lambda: uff_terms(pos, groups[:n_base])}[kind]()
    if kind in (MMFF, UFF) and len(groups) > n_base:
        parts = list(parts) + constraint_terms(pos[:, :3], groups[n_base:])
    if per_group:
        return [float(p.sum()) for p in parts]
    return float(sum(p.sum() for p in parts))


def system_gradient(kind: int, pos: np.ndarray, groups, w0: float = 1.0, w1: float = 1.0, coord_start: int = 0,
                    h: float = 1e-5) -> np.ndarray:
    """Central finite differences of system_energy, with RDKit's half-gradient convention for the chiral and
    fourth-dimension terms of the DG field (dist_geom_kernels_device.cuh:172-176, :229) and the reference's sign
    convention for the C2 part of the UFF inversion gradient (uff_kernels_device.cuh:497)."""
    def fd(energy_fn):
        g = np.zeros_like(pos)
        for a in range(pos.shape[0]):
            for c in range(pos.shape[1]):
                p = pos.copy()
                p[a, c] += h
                ep = energy_fn(p)
                p[a, c] -= 2 * h
                g[a, c] = (ep - energy_fn(p)) / (2 * h)
        return g

    if kind == UFF:  # the inversion gradient follows the reference's sign convention for its C2 part (see uff_terms)
        nb = len(LAYOUT[UFF])
        return fd(lambda p: float(sum(t.sum() for t in uff_terms(p, groups[:nb], gradient_convention=True)) +
                                  sum(t.sum() for t in constraint_terms(p, groups[nb:]))))
    if kind != DG:
        return fd(lambda p: system_energy(kind, p, groups, w0, w1, coord_start))
    empty = [(np.zeros((0, n), dtype=np.int64), np.zeros((0, m))) for n, m in LAYOUT[DG]]
    only = lambda k: [groups[i] if i == k else empty[i] for i in range(3)]  # noqa: E731
    return (fd(lambda p: system_energy(DG, p, only(0), w0, w1)) + 0.5 * fd(lambda p: system_energy(DG, p, only(1), w0, w1)) +
            0.5 * fd(lambda p: system_energy(DG, p, only(2), w0, w1)))


# ---- BFGS (src/minimizer/bfgs_minimize_permol_kernels.cu:35-745, i.e. RDKit BFGSOpt.h) --------------

FUNCTOL, MOVETOL, TOLX, EPS = 1e-4, 1e-7, 4 * 3e-8, 3e-8


def bfgs_minimize(energy, gradient, x0: np.ndarray, max_iters: int = 200, grad_tol: float = 1e-4, scale_grads: bool = True):
    """Returns (x, energy, converged, iterations).  energy(x) -> float, gradient(x) -> array like x (flat)."""
    x = np.array(x0, dtype=np.float64).ravel().copy()
    n = x.size

    def scaled_grad(p):
        g = np.array(gradient(p), dtype=np.float64).ravel()
        scale = 0.1 if scale_grads else 1.0
        if scale_grads:
            g = g * scale
        mx = np.abs(g).max() if n else 0.0
        if scale_grads and mx > 10.0:
            while mx * scale > 10.0:
                scale *= 0.5
            g = g * scale
        return g, scale

    e_prev = energy(x)
    g, gscale = scaled_grad(x)
    d = -g
    hinv = np.eye(n)
    max_step2 = 1e4 * max(float((x * x).sum()), float(n) * n)
    converged = False
    it = 0
    while not converged and it < max_iters:
        old = x.copy()
        s = float((d * d).sum())
        if s > max_step2:
            d = d * np.sqrt(max_step2 / s)
        slope = float((d * g).sum())
        test = float((np.abs(d) / np.maximum(np.abs(x), 1.0)).max())
        lam_min = MOVETOL / (test if test > 0 else 1e-20)
        lam, lam2, e2, e_new = 1.0, 0.0, 0.0, e_prev
        trial = old
        for ls in range(1000):
            trial = old + lam * d
            e_new = energy(trial)
            e_diff = e_new - e_prev
            if lam < lam_min or e_diff <= FUNCTOL * lam * slope:
                break
            if ls == 0:
                tmp = -slope / (2.0 * (e_diff - slope))
            else:
                rhs1 = e_diff - lam * slope
                rhs2 = e2 - e_prev - lam2 * slope
                a = (rhs1 / lam**2 - rhs2 / lam2**2) / (lam - lam2)
                b = (-lam2 * rhs1 / lam**2 + lam * rhs2 / lam2**2) / (lam - lam2)
                if a == 0.0:
                    tmp = -slope / (2.0 * b)
                else:
                    disc = b * b - 3.0 * a * slope
                    if disc < 0.0:
                        tmp = 0.5 * lam
                    elif b <= 0.0:
                        tmp = (-b + np.sqrt(disc)) / (3.0 * a)
                    else:
                        tmp = -slope / (b + np.sqrt(disc))
                tmp = min(tmp, 0.5 * lam)
            lam2, e2 = lam, e_new
            lam = max(tmp, 0.1 * lam)
        x = trial
        xi = x - old
        e_prev = e_new
        if float((np.abs(xi) / np.maximum(np.abs(x), 1.0)).max()) < TOLX:
            converged = True
            break
        g_old = g
        g, gscale = scaled_grad(x)
        dg = g - g_old
        if float((np.abs(g) * np.maximum(np.abs(x), 1.0)).max()) / max(e_prev * gscale, 1.0) < grad_tol:
            converged = True
            break
        hinv, _, _, d = inverse_hessian_update(hinv, dg, xi, g)
        it += 1
    return x.reshape(np.shape(x0)), e_prev, converged, it


def inverse_hessian_update(hinv: np.ndarray, dgrad: np.ndarray, xi: np.ndarray, grad: np.ndarray):
    """One BFGS update of the inverse Hessian and the next search direction, as a step of its own (the operation the reference
    tests in isolation: updateInverseHessianBFGSBatch, src/minimizer/bfgs_hessian.cu, against the loop of
    tests/test_bfgs_hessian.cpp:27-78, i.e. RDKit's BFGSOpt.h): hessDGrad = H dGrad; when dGrad . xi > sqrt(EPS |dGrad|^2 |xi|^2)
    the rank-two update H += xi xi^T / fac - hdg hdg^T / fae + fae u u^T with u = xi / fac - hdg / fae, which also replaces
    dGrad; then xi = -H grad.  Returns (H, hessDGrad, dGrad, xi); the arguments are left alone."""
    hinv, dgrad, xi, grad = (np.array(a, dtype=np.float64) for a in (hinv, dgrad, xi, grad))
    hdg = hinv @ dgrad
    fac, fae = float(dgrad @ xi), float(dgrad @ hdg)
    if fac > 0 and fac * fac > EPS * float(dgrad @ dgrad) * float(xi @ xi):
        fac, fad = 1.0 / fac, 1.0 / fae
        dgrad = fac * xi - fad * hdg
        hinv = hinv + fac * np.outer(xi, xi) - fad * np.outer(hdg, hdg) + fae * np.outer(dgrad, dgrad)
    return hinv, hdg, dgrad, -(hinv @ grad)
