"""Known answers for the stereo-check oracle (oracle/stereo.py), computed by hand from the geometry."""
import math

import numpy as np

from oracle import stereo as S

TET = np.array([[0.0, 0.0, 0.0], [1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], dtype=float)  # centre + regular tetrahedron


def test_regular_tetrahedron_passes_and_planar_centre_fails():
    ix = [0, 1, 2, 3, 4]
    # unit vectors of a regular tetrahedron: |a . (b x c)| = 4 / (3 sqrt 3) = 0.770 > 0.5
    assert not S.fails(S.TETRAHEDRAL, TET, ix, [0.0, 0.0])
    assert not S.fails(S.CHIRAL_CENTER_VOLUME, TET, ix, [0.0, 0.0])
    flat = np.array([[0.0, 0, 0], [1, 0, 0], [-0.5, 0.9, 0], [-0.5, -0.9, 0], [0.1, 0.1, 1.0]])
    assert S.fails(S.TETRAHEDRAL, flat, ix, [0.0, 0.0])          # neighbours 1-3 and the centre are coplanar: volume 0
    squashed = TET.copy()
    squashed[1:, 2] *= 0.3                                       # triple products 4 z / (2 + z^2)^1.5 = 0.397: fails at 0.5, passes at 0.125
    assert S.fails(S.TETRAHEDRAL, squashed, ix, [0.0, 0.0])
    assert not S.fails(S.TETRAHEDRAL, squashed, ix, [1.0, 0.0])  # in fused small rings the limit is 0.25 * 0.5


def test_centre_outside_the_neighbour_volume_fails_only_the_volume_part():
    out = TET.copy()
    out[0] = [3.0, 3.0, 3.0]                                     # beyond vertex 1, outside the tetrahedron
    assert S.fails(S.CHIRAL_CENTER_VOLUME, out, [0, 1, 2, 3, 4], [0.0, 0.0])
    three = [0, 1, 2, 3, 0]                                      # three-coordinate centre: idx4 == idx0, no volume test
    assert not S.fails(S.CHIRAL_CENTER_VOLUME, out, three, [0.0, 0.0])
    near = TET.copy()
    near[0] = [0.99, 0.99, 0.99]   # inside, but its unnormalised plane products with the faces through vertex 1 are 0.04 < tol 0.1
    assert S.fails(S.CHIRAL_CENTER_VOLUME, near, [0, 1, 2, 3, 4], [0.0, 0.0])


def test_chiral_volume_window():
    # V = (p1 - p4) . ((p2 - p4) x (p3 - p4)) with p4 the origin and the unit axes: V = 1
    p = np.array([[9.0, 9, 9], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 0, 0]])
    ix = [0, 1, 2, 3, 4]
    assert not S.fails(S.CHIRAL_VOLUME, p, ix, [0.9, 1.5])    # inside [lb, ub]
    assert not S.fails(S.CHIRAL_VOLUME, p, ix, [1.2, 2.0])    # below lb but V / lb = 0.83 >= 0.8
    assert S.fails(S.CHIRAL_VOLUME, p, ix, [1.3, 2.0])        # V / lb = 0.77 < 0.8
    q = p.copy()
    q[[1, 2]] = q[[2, 1]]                                     # swap two neighbours: V = -1
    assert S.fails(S.CHIRAL_VOLUME, q, ix, [0.9, 1.5])        # wrong sign against a positive lower bound
    assert not S.fails(S.CHIRAL_VOLUME, q, ix, [-1.5, -0.9])  # inside the negative window
    assert S.fails(S.CHIRAL_VOLUME, p, ix, [-1.5, -0.9])      # +1 against a negative upper bound


def test_chiral_distance_window():
    p = np.array([[0.0, 0, 0], [2.0, 0, 0]])
    ix = [0, 1, 0, 0, 0]
    assert not S.fails(S.CHIRAL_DISTANCE, p, ix, [1.5, 2.5])
    assert not S.fails(S.CHIRAL_DISTANCE, p, ix, [2.1, 3.0])   # 0.1 below lb, slack 0.1 * ub = 0.3
    assert S.fails(S.CHIRAL_DISTANCE, p, ix, [2.5, 3.0])       # 0.5 below lb > 0.3
    assert not S.fails(S.CHIRAL_DISTANCE, p, ix, [1.0, 1.9])   # 0.1 above ub, slack 0.19
    assert S.fails(S.CHIRAL_DISTANCE, p, ix, [1.0, 1.5])       # 0.5 above ub > 0.15


def test_double_bond_stereo_and_geometry():
    cis = np.array([[-1.0, 1, 0], [0, 0, 0], [1.5, 0, 0], [2.5, 1, 0]])       # dihedral 0
    trans = np.array([[-1.0, 1, 0], [0, 0, 0], [1.5, 0, 0], [2.5, -1, 0]])    # dihedral pi
    ix = [0, 1, 2, 3, 0]
    assert S.fails(S.DOUBLE_BOND_STEREO, cis, ix, [1.0, 0.0])        # sign +1 wants angle > pi/2 (trans)
    assert not S.fails(S.DOUBLE_BOND_STEREO, cis, ix, [-1.0, 0.0])
    assert not S.fails(S.DOUBLE_BOND_STEREO, trans, ix, [1.0, 0.0])
    assert S.fails(S.DOUBLE_BOND_STEREO, trans, ix, [-1.0, 0.0])
    bent = np.array([[-1.0, 0.5, 0], [0, 0, 0], [1, 0, 0]])
    line = np.array([[-1.0, 0.0, 0], [0, 0, 0], [1, 0, 0]])
    almost = np.array([[-1.0, 0.05, 0], [0, 0, 0], [1, 0, 0]])                # cos = -0.99875: 1 + cos = 1.25e-3 > 1e-3
    gx = [0, 1, 2, 0, 0]
    assert not S.fails(S.DOUBLE_BOND_GEOMETRY, bent, gx, [0.0, 0.0])
    assert S.fails(S.DOUBLE_BOND_GEOMETRY, line, gx, [0.0, 0.0])
    assert not S.fails(S.DOUBLE_BOND_GEOMETRY, almost, gx, [0.0, 0.0])
    assert math.isclose(1.0 - 1.0 / math.sqrt(1.0 + 0.05 ** 2), 1.2476e-3, rel_tol=1e-3)
