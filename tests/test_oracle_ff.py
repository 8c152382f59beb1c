"""Pins the force-field / BFGS oracle (oracle/ff.py) without RDKit: closed-form single terms and the reference's
RDKit-free quartic BFGS case (tests/test_bfgs_minimizer.cu:823-1029, :1139-1147)."""

import numpy as np
import pytest

from oracle import ff
from tests import util


def test_dg_terms_closed_form():
    pos = np.array([[0.0, 0, 0, 0], [2.0, 0, 0, 0], [0, 1.0, 0, 0.5], [0, 0, 1.0, 0]])
    e = np.zeros((0, 2), dtype=int)
    inside = [(np.array([[0, 1]]), np.array([[1.0, 9.0, 1.0]])), (np.zeros((0, 4), int), np.zeros((0, 2))), (np.zeros((0, 1), int), np.zeros((0, 0)))]
    assert ff.system_energy(ff.DG, pos, inside) == 0.0                       # lb2 <= d2 = 4 <= ub2
    over = [(np.array([[0, 1]]), np.array([[1.0, 2.0, 3.0]])), inside[1], inside[2]]
    assert ff.system_energy(ff.DG, pos, over) == pytest.approx(3.0 * (4.0 / 2.0 - 1.0) ** 2)
    under = [(np.array([[0, 1]]), np.array([[16.0, 25.0, 1.0]])), inside[1], inside[2]]
    assert ff.system_energy(ff.DG, pos, under) == pytest.approx((2 * 16.0 / (16.0 + 4.0) - 1.0) ** 2)
    # chiral volume of (e_x*2, e_y, e_z) about the origin = 2; bounds [3, 4] -> w (2 - 3)^2
    chiral = [(e, np.zeros((0, 3))), (np.array([[1, 2, 3, 0]]), np.array([[3.0, 4.0]])), inside[2]]
    assert ff.system_energy(ff.DG, pos, chiral, w0=0.2) == pytest.approx(0.2)
    fourth = [(e, np.zeros((0, 3))), inside[1], (np.array([[2]]), np.zeros((1, 0)))]
    assert ff.system_energy(ff.DG, pos, fourth, w1=0.1) == pytest.approx(0.1 * 0.25)


def test_mmff_terms_closed_form():
    pos = np.array([[0.0, 0, 0], [1.5, 0, 0], [1.5, 1.2, 0], [3.0, 1.2, 0.0]])
    empty = lambda n, m: (np.zeros((0, n), int), np.zeros((0, m)))  # noqa: E731
    groups = [empty(n, m) for n, m in ff.LAYOUT[ff.MMFF]]
    g = list(groups)
    g[0] = (np.array([[0, 1]]), np.array([[1.5, 5.0]]))
    assert ff.system_energy(ff.MMFF, pos, g) == 0.0                          # bond at rest length
    g[0] = (np.array([[0, 1]]), np.array([[1.4, 5.0]]))
    dr = 0.1
    assert ff.system_energy(ff.MMFF, pos, g) == pytest.approx(143.9325 / 2 * 5.0 * dr**2 * (1 - 2 * dr + 7 / 12 * 4 * dr**2))
    g = list(groups)
    g[1] = (np.array([[0, 1, 2]]), np.array([[90.0, 0.8, 0.0]]))
    assert ff.system_energy(ff.MMFF, pos, g) == pytest.approx(0.0, abs=1e-20)  # right angle at rest
    g[1] = (np.array([[0, 1, 2]]), np.array([[90.0, 0.8, 1.0]]))
    assert ff.system_energy(ff.MMFF, pos, g) == pytest.approx(143.9325 * 0.8)  # linear form, cos = 0
    g = list(groups)
    g[4] = (np.array([[0, 1, 2, 3]]), np.array([[1.0, 2.0, 3.0]]))           # trans dihedral: cos(phi) = -1
    assert ff.system_energy(ff.MMFF, pos, g) == pytest.approx(0.5 * (0 + 0 + 0))
    g = list(groups)
    g[6] = (np.array([[0, 1]]), np.array([[0.25, 1.0, 1.0]]))
    assert ff.system_energy(ff.MMFF, pos, g) == pytest.approx(0.75 * 332.0716 * 0.25 / 1.55)


def test_mmff_terms_equal_the_published_functional_forms():
    """MMFF94 as published (Halgren, J. Comput. Chem. 17, 490-519 (1996), eqs. 2-8) with the constants RDKit's MMFF code
    documents: angle bending 0.043844 ka/2 dt^2 (1 + cb dt) with cb = -0.006981317 / degree, stretch-bend 2.51210 (kba_ijk dr_ij
    + kba_kji dr_kj) dt, out-of-plane 0.043844 koop/2 chi^2, torsion 0.5 (V1 (1 + cos p) + V2 (1 - cos 2p) + V3 (1 + cos 3p)),
    buffered 14-7 van der Waals, buffered Coulomb 332.0716 q_i q_j / (D (R + 0.05)).  Every expected value below is written
    out from those formulas with the geometry computed by hand — none goes through oracle/ff.py's helpers."""
    empty = lambda n, m: (np.zeros((0, n), int), np.zeros((0, m)))  # noqa: E731
    groups = [empty(n, m) for n, m in ff.LAYOUT[ff.MMFF]]
    # a bent three-atom chain: |r01| = 1.1, |r21| = 1.4, angle 100 degrees at atom 1
    th = np.deg2rad(100.0)
    pos = np.array([[1.1, 0.0, 0.0], [0.0, 0.0, 0.0], [1.4 * np.cos(th), 1.4 * np.sin(th), 0.0], [0.3, 0.4, 1.2]])
    g = list(groups)
    g[1] = (np.array([[0, 1, 2]]), np.array([[109.5, 0.75, 0.0]]))
    dt = 100.0 - 109.5
    assert ff.system_energy(ff.MMFF, pos, g) == pytest.approx(0.043844 * 0.75 / 2 * dt**2 * (1 - 0.006981317 * dt), rel=2e-5)  # 0.043844 is the published 5-digit value of 143.9325 (pi / 180)^2
    g = list(groups)
    g[2] = (np.array([[0, 1, 2]]), np.array([[109.5, 1.0, 1.5, 0.3, 0.2]]))   # theta0, r0_ij, r0_kj, kba_ijk, kba_kji
    assert ff.system_energy(ff.MMFF, pos, g) == pytest.approx(2.51210 * dt * (0.3 * (1.1 - 1.0) + 0.2 * (1.4 - 1.5)), rel=1e-9)
    # out of plane: atoms 0, 2 and the centre 1 span the xy plane, atom 3 sits at (0.3, 0.4, 1.2): sin(chi) = z / |r| = 1.2 / 1.3
    g = list(groups)
    g[3] = (np.array([[0, 1, 2, 3]]), np.array([[0.05]]))
    chi = np.rad2deg(np.arcsin(1.2 / 1.3))
    assert ff.system_energy(ff.MMFF, pos, g) == pytest.approx(0.043844 * 0.05 / 2 * chi**2, rel=2e-5)
    # torsion 0-1-2-3 of 60 degrees: cos p = 0.5, cos 2p = -0.5, cos 3p = -1
    ph = np.deg2rad(60.0)
    ptor = np.array([[1.0, 1.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 0.0], [np.cos(ph), 0.0, np.sin(ph)]])
    ptor[0] = [np.cos(0.0), 1.0, np.sin(0.0)]
    g = list(groups)
    g[4] = (np.array([[0, 1, 2, 3]]), np.array([[0.7, -1.1, 0.4]]))
    assert ff.system_energy(ff.MMFF, ptor, g) == pytest.approx(0.5 * (0.7 * 1.5 + -1.1 * 1.5 + 0.4 * 0.0), rel=1e-12)
    # buffered 14-7 at R = 3.2 with R* = 3.6, eps = 0.07
    pv = np.array([[0.0, 0.0, 0.0], [3.2, 0.0, 0.0]])
    g = list(groups)
    g[5] = (np.array([[0, 1]]), np.array([[3.6, 0.07]]))
    want = 0.07 * (1.07 * 3.6 / (3.2 + 0.07 * 3.6)) ** 7 * (1.12 * 3.6**7 / (3.2**7 + 0.12 * 3.6**7) - 2.0)
    assert ff.system_energy(ff.MMFF, pv, g) == pytest.approx(want, rel=1e-12)
    assert 0.0 < want < 0.01                                                          # on the repulsive wall, just past the zero crossing
    g[5] = (np.array([[0, 1]]), np.array([[3.2, 0.07]]))
    assert ff.system_energy(ff.MMFF, pv, g) == pytest.approx(-0.07, rel=1e-12)      # the minimum: exactly -eps at R = R*
    # buffered Coulomb, distance-dependent dielectric (model 2) and the 0.75 scaling of 1-4 pairs
    g = list(groups)
    g[6] = (np.array([[0, 1]]), np.array([[-0.18, 2.0, 0.0]]))
    assert ff.system_energy(ff.MMFF, pv, g) == pytest.approx(332.0716 * -0.18 / 3.25**2, rel=1e-12)
    g[6] = (np.array([[0, 1]]), np.array([[-0.18, 1.0, 1.0]]))
    assert ff.system_energy(ff.MMFF, pv, g) == pytest.approx(0.75 * 332.0716 * -0.18 / 3.25, rel=1e-12)


def test_uff_terms_closed_form():
    """Hand-computable UFF values (functional forms of reference src/forcefields/uff_kernels_device.cuh:37-580)."""
    pos = np.array([[0.0, 0, 0], [1.5, 0, 0], [1.5, 1.2, 0], [1.5, 1.2, 1.0]])
    empty = lambda n, m: (np.zeros((0, n), int), np.zeros((0, m)))  # noqa: E731
    groups = [empty(n, m) for n, m in ff.LAYOUT[ff.UFF]]
    g = list(groups)
    g[0] = (np.array([[0, 1]]), np.array([[1.3, 700.0]]))
    assert ff.system_energy(ff.UFF, pos, g) == pytest.approx(0.5 * 700.0 * 0.2**2)
    # right angle 0-1-2: cos = 0, cos 2t = -1, cos 3t = 0, cos 4t = 1
    for order, want in [(0, 100.0 * (0.3 + 0.0 - 0.2)), (1, 100.0 * 1.0), (2, 100.0 * 2.0 / 4.0), (3, 100.0 / 9.0), (4, 0.0)]:
        g = list(groups)
        g[1] = (np.array([[0, 1, 2]]), np.array([[1.9, 100.0, order, 0.3, 0.5, 0.2]]))
        assert ff.system_energy(ff.UFF, pos, g) == pytest.approx(want, abs=1e-12)
    # 20 degree angle (cos > 0.866) with order 2: base term + exp(-20 (theta - theta0 + 0.25))
    th = np.deg2rad(20.0)
    p20 = np.array([[np.cos(th), np.sin(th), 0.0], [0, 0, 0], [1.0, 0, 0]])
    g = list(groups)
    g[1] = (np.array([[0, 1, 2]]), np.array([[0.5, 80.0, 2, 0, 0, 0]]))
    assert ff.system_energy(ff.UFF, p20, g) == pytest.approx(80.0 * (1 - np.cos(2 * th)) / 4 + np.exp(-20 * (th - 0.5 + 0.25)))
    g[1] = (np.array([[0, 1, 2]]), np.array([[0.5, 80.0, 0, 1.0, 2.0, 3.0]]))  # order 0 never gets the correction
    assert ff.system_energy(ff.UFF, p20, g) == pytest.approx(80.0 * (1 + 2 * np.cos(th) + 3 * np.cos(2 * th)))
    # dihedral 0-1-2-3 is 90 degrees: cos 2p = -1, cos 3p = 0, cos 6p = -1
    for order, cosn in [(2, -1.0), (3, 0.0), (6, -1.0), (4, None)]:
        g = list(groups)
        g[2] = (np.array([[0, 1, 2, 3]]), np.array([[7.0, order, -1.0]]))
        assert ff.system_energy(ff.UFF, pos, g) == pytest.approx(0.0 if cosn is None else 3.5 * (1 + cosn), abs=1e-12)
    # inversion: apex 3 perpendicular to the plane (0, 1, 2) about centre... use centre 2 with arms 1, 3 and apex 0
    pinv = np.array([[0.0, 0, 0], [1.0, 0, 0], [0, 1.0, 0], [0.3, 0.3, 0.0]])   # planar: cosY = 0, sinY = 1
    g = list(groups)
    g[3] = (np.array([[1, 0, 2, 3]]), np.array([[6.0, 1.0, -1.0, 0.0]]))
    assert ff.system_energy(ff.UFF, pinv, g) == pytest.approx(0.0, abs=1e-12)     # sp2 centre in its plane
    g[3] = (np.array([[1, 0, 2, 3]]), np.array([[6.0, 0.2, 0.3, 0.4]]))
    assert ff.system_energy(ff.UFF, pinv, g) == pytest.approx(6.0 * (0.2 + 0.3 + 0.4))
    # 12-6: minimum -D at r = x_ij, zero beyond the threshold
    g = list(groups)
    g[4] = (np.array([[0, 1]]), np.array([[1.5, 0.2, 10.0]]))
    assert ff.system_energy(ff.UFF, pos, g) == pytest.approx(-0.2)
    g[4] = (np.array([[0, 1]]), np.array([[1.5, 0.2, 1.4]]))
    assert ff.system_energy(ff.UFF, pos, g) == 0.0


def test_uff_inversion_gradient_convention_matches_reference_formula():
    """The reference's analytic inversion gradient (restated in oracle/ff.py) equals the finite difference of the
    energy with the C2 part NEGATED, and differs from the derivative of the reported energy when C2 != 0."""
    rng = np.random.default_rng(3)
    empty = lambda n, m: (np.zeros((0, n), int), np.zeros((0, m)))  # noqa: E731
    for c0, c1, c2 in [(1.0, -1.0, 0.0), (0.3, -0.7, 1.0), (0.0, 0.0, 1.0)]:
        pos = rng.normal(size=(4, 3))
        idx, par = np.array([0, 1, 2, 3]), np.array([2.5, c0, c1, c2])
        groups = [empty(n, m) for n, m in ff.LAYOUT[ff.UFF]]
        groups[3] = (idx[None, :], par[None, :])
        analytic = ff.uff_inversion_gradient_reference(pos, idx, par)
        conv = ff.system_gradient(ff.UFF, pos, groups, h=1e-6)
        assert np.allclose(analytic, conv, rtol=1e-6, atol=1e-7)
        true = np.zeros_like(pos)
        for a in range(4):
            for c in range(3):
                q = pos.copy()
                q[a, c] += 1e-6
                ep = ff.system_energy(ff.UFF, q, groups)
                q[a, c] -= 2e-6
                true[a, c] = (ep - ff.system_energy(ff.UFF, q, groups)) / 2e-6
        assert np.allclose(analytic, true, rtol=1e-5, atol=1e-6) == (c2 == 0.0)


def test_etk_terms_closed_form():
    pos = np.zeros((4, 4))
    pos[:, :3] = [[0, 0, 0], [1.5, 0, 0], [1.5, 1.2, 0], [1.5, 1.2, 1.0]]    # dihedral 90 deg -> cos = 0
    empty = lambda n, m: (np.zeros((0, n), int), np.zeros((0, m)))  # noqa: E731
    groups = [empty(n, m) for n, m in ff.LAYOUT[ff.ETK]]
    g = list(groups)
    fc = np.array([[1.0, 2, 3, 4, 5, 6, 1, 1, 1, 1, 1, 1]])
    g[0] = (np.array([[0, 1, 2, 3]]), fc)
    # cos k*90deg = 0, -1, 0, 1, 0, -1
    assert ff.system_energy(ff.ETK, pos, g) == pytest.approx(1 + 2 * 0 + 3 + 4 * 2 + 5 + 6 * 0)
    g = list(groups)
    g[2] = (np.array([[0, 1]]), np.array([[1.0, 1.2, 100.0, 0.0]]))
    assert ff.system_energy(ff.ETK, pos, g) == pytest.approx(0.5 * 100 * 0.3**2)
    g[2] = (np.array([[0, 1]]), np.array([[1.0, 2.0, 100.0, 0.0]]))
    assert ff.system_energy(ff.ETK, pos, g) == 0.0
    g = list(groups)
    g[4] = (np.array([[0, 1, 2]]), np.array([[100.0, 120.0]]))
    assert ff.system_energy(ff.ETK, pos, g) == pytest.approx(10.0**2)


@pytest.mark.parametrize("kind", [ff.DG, ff.ETK, ff.MMFF, ff.UFF])
def test_finite_difference_gradient_is_self_consistent(kind):
    rng = np.random.default_rng(kind)
    pos, groups = util.random_ff_system(kind, 9, rng)
    g1 = ff.system_gradient(kind, pos, groups, 0.7, 0.3, h=1e-5)
    g2 = ff.system_gradient(kind, pos, groups, 0.7, 0.3, h=2e-5)
    assert np.allclose(g1, g2, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("last_dim", [True, False])
def test_bfgs_quartic_known_answer(last_dim):
    # systems atomStarts = {0, 3, 10, 12}, dim 4, start p + U(-2, 2); converge to x_p = p within 0.1
    starts = [0, 3, 10, 12]
    rng = np.random.default_rng(42)
    for s in range(3):
        n = (starts[s + 1] - starts[s]) * 4
        c0 = starts[s] * 4
        x0 = c0 + np.arange(n) + rng.uniform(-2, 2, size=n)
        shape = (n // 4, 4)
        w0 = 1.0 if last_dim else 0.0
        e = lambda x: ff.system_energy(ff.QUARTIC, x.reshape(shape), [], w0, 0.0, c0)  # noqa: E731

        def g(x):
            d = x - (c0 + np.arange(n))
            gr = 4 * d**3
            if not last_dim:
                gr.reshape(shape)[:, 3] = 0.0
            return gr

        x, energy, conv, iters = ff.bfgs_minimize(e, g, x0, max_iters=400, grad_tol=1e-5, scale_grads=False)
        target = c0 + np.arange(n)
        mask = np.ones(n, bool) if last_dim else (np.arange(n) % 4 != 3)
        assert np.abs(x - target)[mask].max() < 0.1
        assert energy < 1e-3
