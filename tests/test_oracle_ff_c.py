"""Pins oracle/oracle_ff.c (C + OpenMP, hand-derived gradients, RDKit BFGS, ETKDG pipeline) against oracle/ff.py
(numpy energies, central finite differences, plain-Python BFGS) and against the reference's RDKit-free BFGS known answer
(tests/test_bfgs_minimizer.cu:823-1029) — before the C oracle is used as a checker or a CPU baseline anywhere else."""

import numpy as np
import pytest

from nvmolkit_amd import synthetic
from oracle import ff as off
from oracle import ffc

KINDS = [off.DG, off.ETK, off.MMFF, off.UFF]


def _systems(kind, sizes, seed):
    rng = np.random.default_rng(seed)
    return [synthetic.random_ff_system(kind, n, rng) for n in sizes]


@pytest.mark.parametrize("kind", KINDS)
def test_energy_matches_numpy_oracle(kind):
    systems = _systems(kind, [1, 2, 3, 4, 7, 13, 24, 40], 11 + kind)
    batch, flat = ffc.batch_from_systems(kind, systems)
    got = batch.energy(flat, 0.7, 0.3)
    want = np.array([off.system_energy(kind, p, g, 0.7, 0.3) for p, g in systems])
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("kind", KINDS)
def test_gradient_matches_finite_differences(kind):
    systems = _systems(kind, [2, 3, 4, 5, 9, 16, 28], 23 + kind)
    batch, flat = ffc.batch_from_systems(kind, systems)
    got = batch.gradient(flat, 0.7, 0.3)
    want = np.concatenate([off.system_gradient(kind, p, g, 0.7, 0.3).reshape(-1) for p, g in systems])
    scale = np.maximum(np.abs(want), 1.0)
    assert np.max(np.abs(got - want) / scale) < 5e-6


def test_group_mask_and_active_mask():
    systems = _systems(off.ETK, [12, 15], 5)
    a_s, flat, groups = synthetic.build_ff_batch_arrays(off.ETK, systems)
    only_imp = ffc.Batch(off.ETK, a_s, groups, group_mask=0x2)
    want = [off.system_energy(off.ETK, p, g, per_group=True)[1] for p, g in systems]
    np.testing.assert_allclose(only_imp.energy(flat), want, rtol=1e-12, atol=1e-12)
    full = ffc.Batch(off.ETK, a_s, groups)
    e = full.energy(flat, active=[0, 1])
    assert e[0] == 0.0 and e[1] != 0.0


def test_system_mol_shares_tables():
    rng = np.random.default_rng(3)
    mols = [synthetic.random_ff_system(off.MMFF, n, rng) for n in (9, 14)]
    from nvmolkit_amd.forcefield import stack_molecule_tables

    groups = stack_molecule_tables(off.MMFF, [m[1] for m in mols])
    sys_mol = [1, 0, 1]
    pos = [mols[m][0] + rng.normal(scale=0.05, size=mols[m][0].shape) for m in sys_mol]
    a_s = np.concatenate([[0], np.cumsum([len(p) for p in pos])])
    b = ffc.Batch(off.MMFF, a_s, groups, system_mol=sys_mol)
    got = b.energy(np.concatenate([p.reshape(-1) for p in pos]))
    want = [off.system_energy(off.MMFF, p, mols[m][1]) for p, m in zip(pos, sys_mol)]
    np.testing.assert_allclose(got, want, rtol=1e-12)


def test_bfgs_quartic_known_answer():
    """The reference's RDKit-free case: systems atomStarts = {0, 3, 10, 12}, dim 4, start i + U(-2, 2) -> x_p = p."""
    a_s = np.array([0, 3, 10, 12], dtype=np.int32)
    rng = np.random.default_rng(42)
    x0 = np.arange(48, dtype=np.float64) + rng.uniform(-2, 2, size=48)
    b = ffc.Batch(off.QUARTIC, a_s, [])
    x, e, st, it = b.minimize(x0, max_iters=400, grad_tol=1e-6, scale_grads=False, w0=1.0)
    assert np.all(st == 0)
    assert np.max(np.abs(x - np.arange(48))) < 0.1


@pytest.mark.parametrize("kind", [off.DG, off.MMFF, off.UFF])
@pytest.mark.parametrize("iters,tol", [(1, 1e-7), (3, 1e-6), (8, 1e-4)])
def test_bfgs_trajectory_matches_python_bfgs(kind, iters, tol):
    """Same algorithm, same field, independent gradients (analytic vs finite difference): the iterates after a fixed
    number of BFGS iterations agree (end points of long runs may legitimately differ — several minima)."""
    systems = _systems(kind, [5, 8, 11], 77 + kind)
    batch, flat = ffc.batch_from_systems(kind, systems)
    x, e, st, it = batch.minimize(flat, max_iters=iters, grad_tol=1e-12)
    o = 0
    for s, (p, g) in enumerate(systems):
        xr, er, conv, itr = off.bfgs_minimize(lambda q: off.system_energy(kind, q.reshape(p.shape), g),
                                              lambda q: off.system_gradient(kind, q.reshape(p.shape), g).reshape(-1), p,
                                              max_iters=iters, grad_tol=1e-12)
        assert itr == it[s]
        np.testing.assert_allclose(x[o:o + p.size], xr.reshape(-1), atol=tol, rtol=0)
        assert abs(e[s] - er) <= 10 * tol * max(1.0, abs(er))
        o += p.size


def test_etkdg_pipeline_embeds_feasible_molecules():
    from nvmolkit_amd.embedMolecules import FlatMolecule

    rng = np.random.default_rng(8)
    built = [synthetic.synthetic_embed_molecule(rng, n) for n in (8, 12, 17)]
    mols = [FlatMolecule(**b[0]) for b in built]
    coords, counts, slots, fails, iters = ffc.etkdg_embed(mols, confs_per_molecule=3, max_iterations=10, enforce_chirality=False,
                                                          seed=5)
    assert np.all(counts == 3) and iters > 0
    for m, (fields, ref, (pairs, lb, ub)) in enumerate(built):
        n = fields["n_atoms"]
        c = coords[slots[m]:slots[m] + 3 * n * counts[m]].reshape(counts[m], n, 3)
        d = np.linalg.norm(c[:, pairs[:, 0]] - c[:, pairs[:, 1]], axis=2)
        viol = np.maximum(np.maximum(lb - d, d - ub), 0.0) / ub
        assert viol.max() < 0.2
    again = ffc.etkdg_embed(mols, confs_per_molecule=3, max_iterations=10, enforce_chirality=False, seed=5)
    assert np.array_equal(coords, again[0])  # one thread per attempt, no shared accumulation: deterministic


def test_random_coords_range_and_moments():
    x = ffc.random_coords(1234, 7, 20000, 10.0)
    assert x.min() >= -5.0 and x.max() < 5.0
    assert abs(x.mean()) < 0.05 and abs(x.var() - 100.0 / 12.0) < 0.15
