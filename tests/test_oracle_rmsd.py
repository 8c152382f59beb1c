"""Pins of the RMSD / pruning oracle (oracle/rmsd.py) by closed forms — no reference output is available without RDKit.
Reference behaviour being restated: src/conformer_rmsd.cu:133-258, rdkit_extensions/conformer_pruning.cpp:88-137,
nvmolkit/tests/test_conformer_rmsd.py (identical / rotated / translated conformers, condensed index order)."""

import numpy as np
import pytest

from oracle import rmsd


def rotation(rng):
    q, r = np.linalg.qr(rng.normal(size=(3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def test_rigid_motions_give_zero_and_mirror_images_do_not():
    rng = np.random.default_rng(1)
    a = rng.normal(size=(23, 3))
    moved = a @ rotation(rng).T + rng.normal(size=3)
    assert rmsd.pair_rmsd(a, a) == pytest.approx(0.0, abs=1e-7)
    assert rmsd.pair_rmsd(a, moved) == pytest.approx(0.0, abs=1e-7)
    assert rmsd.pair_rmsd(a, moved, prealigned=True) > 0.5            # raw coordinates differ
    mirror = a * np.array([1.0, 1.0, -1.0])
    assert rmsd.pair_rmsd(a, mirror) > 0.1                             # proper rotations only: no reflection allowed
    planar = a.copy()
    planar[:, 2] = 0.0
    assert rmsd.pair_rmsd(planar, planar * np.array([1.0, 1.0, -1.0])) == pytest.approx(0.0, abs=1e-7)


def test_known_displacements():
    # two atoms on the x axis at +-1 vs +-2: already optimally aligned, every atom is displaced by 1
    a = np.array([[-1.0, 0, 0], [1.0, 0, 0]])
    b = np.array([[-2.0, 0, 0], [2.0, 0, 0]])
    assert rmsd.pair_rmsd(a, b) == pytest.approx(1.0)
    assert rmsd.pair_rmsd(a, b, prealigned=True) == pytest.approx(1.0)
    # prealigned keeps the translation: a shift of (3, 4, 0) is an RMSD of 5
    assert rmsd.pair_rmsd(a, a + np.array([3.0, 4.0, 0.0]), prealigned=True) == pytest.approx(5.0)
    assert rmsd.pair_rmsd(a, a + np.array([3.0, 4.0, 0.0])) == pytest.approx(0.0, abs=1e-7)


def test_condensed_order_and_symmetry():
    rng = np.random.default_rng(2)
    confs = rng.normal(size=(5, 9, 3))
    m = rmsd.rms_matrix(confs)
    assert m.shape == (10,)
    assert m[3 * 2 // 2 + 1] == pytest.approx(rmsd.pair_rmsd(confs[3], confs[1]))      # pair (3, 1) at i (i - 1) / 2 + j
    assert m[4 * 3 // 2 + 0] == pytest.approx(rmsd.pair_rmsd(confs[0], confs[4]))      # symmetric in its arguments
    assert (m >= 0).all()


def test_greedy_pruning():
    rng = np.random.default_rng(3)
    base = rng.normal(size=(12, 3))
    far = rng.normal(size=(12, 3))
    confs = np.stack([base, base @ rotation(rng).T, far, base + 0.01 * rng.normal(size=base.shape), far + 5.0])
    keep = rmsd.prune(confs, 0.1)
    assert keep.tolist() == [True, False, True, False, False]       # rotated copy, noisy copy and translated copy go
    assert rmsd.prune(confs, 0.0).all()
    assert rmsd.prune(confs, 1e9).tolist() == [True, False, False, False, False]


def test_symmetric_rmsd_is_the_smallest_over_the_self_matches():
    """A conformer whose equivalent atoms are swapped is the same conformer once the self matches are searched (reference:
    _isConfFarFromRest loops the matches, rdkit_extensions/conformer_pruning.cpp:101-112)."""
    rng = np.random.default_rng(2)
    a = rng.normal(size=(6, 3)) * 2.0
    swap = np.array([1, 0, 2, 3, 5, 4])
    b = a[swap]                                        # atoms 0 <-> 1 and 4 <-> 5 exchanged: a relabelling, not a new shape
    ident = np.arange(6)
    matches = np.stack([ident, swap])
    assert rmsd.pair_rmsd(a, b) > 0.5
    assert rmsd.pair_rmsd_sym(a, b, matches) == pytest.approx(0.0, abs=1e-7)
    assert rmsd.pair_rmsd_sym(a, b, ident[None]) == pytest.approx(rmsd.pair_rmsd(a, b))
    sub = np.stack([ident[:4], swap[:4]])              # matches over an atom subset (the heavy atoms): RMSD on those atoms only
    assert rmsd.pair_rmsd_sym(a, b, sub) == pytest.approx(min(rmsd.pair_rmsd(a[:4], b[:4]), rmsd.pair_rmsd(a[:4], b[swap[:4]])))
    confs = np.stack([a, b, a + 0.4 * rng.normal(size=a.shape), a[swap] + 1e-3])
    assert rmsd.prune(confs, 0.1).tolist() == [True, True, True, False]      # the last is conformer 1 shifted: plain RMSD sees that much
    assert rmsd.prune_sym(confs, 0.1, matches).tolist() == [True, False, True, False]
    m = rmsd.rms_matrix_sym(confs, matches)
    assert m[0] == pytest.approx(0.0, abs=1e-7) and m[3 * 2 // 2 + 1] < 0.01 and m[1] > 0.1
