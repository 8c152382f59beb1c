"""E6: the stereochemistry check stages through the C ABI against the oracle (oracle/stereo.py), per check kind, on
random systems whose terms sit on both sides of every threshold, plus the hand-made cases of tests/test_oracle_stereo.py."""
import numpy as np
import pytest
import torch

from nvmolkit_amd.embedMolecules import stereo_check_flat
from oracle import stereo as S

pytestmark = pytest.mark.gpu

N_IDX = {S.TETRAHEDRAL: 5, S.CHIRAL_CENTER_VOLUME: 5, S.CHIRAL_VOLUME: 5, S.CHIRAL_DISTANCE: 2, S.DOUBLE_BOND_STEREO: 4,
         S.DOUBLE_BOND_GEOMETRY: 3}


def random_case(rng, n_mols=40, n_sys=300):
    """Molecules with a mixed list of check terms; systems = random geometries of random molecules."""
    mol_atoms = rng.integers(6, 14, size=n_mols)
    starts, kinds, idx, par = [0], [], [], []
    for m in range(n_mols):
        na = int(mol_atoms[m])
        for _ in range(int(rng.integers(3, 9))):
            k = int(rng.integers(0, 6))
            ix = list(rng.choice(na, size=5, replace=False))
            if k in (S.TETRAHEDRAL, S.CHIRAL_CENTER_VOLUME) and rng.random() < 0.3:
                ix[4] = ix[0]                                   # three-coordinate centre
            if k == S.DOUBLE_BOND_GEOMETRY and rng.random() < 0.5:
                ix[:3] = [0, 1, 2]                              # the triple some systems put on a line (below)
            if k == S.CHIRAL_VOLUME:
                lo = rng.uniform(-3, 3)
                p = [lo, lo + rng.uniform(0.1, 2.0)]
            elif k == S.CHIRAL_DISTANCE:
                lo = rng.uniform(0.5, 3.0)
                p = [lo, lo + rng.uniform(0.1, 1.5)]
            elif k == S.DOUBLE_BOND_STEREO:
                p = [float(rng.choice([-1.0, 1.0])), 0.0]
            elif k == S.TETRAHEDRAL:
                p = [float(rng.random() < 0.4), 0.0]            # inFusedSmallRings flag
            else:
                p = [0.0, 0.0]
            kinds.append(k)
            idx.append(ix)
            par.append(p)
        starts.append(len(kinds))
    sys_mol = rng.integers(0, n_mols, size=n_sys)
    atom_starts = np.zeros(n_sys + 1, dtype=np.int64)
    atom_starts[1:] = np.cumsum(mol_atoms[sys_mol])
    pos = rng.normal(scale=1.3, size=(int(atom_starts[-1]), 4))
    # some collinear / coplanar / degenerate-looking geometry so that the boundary branches are taken
    for s in range(0, n_sys, 7):
        a0 = int(atom_starts[s])
        pos[a0 + 2, :3] = 2.0 * pos[a0 + 1, :3] - pos[a0, :3]   # atoms 0, 1, 2 on a line
    for s in range(3, n_sys, 11):
        a0 = int(atom_starts[s])
        pos[a0:a0 + 5, 2] = 0.0                                 # first five atoms in a plane
    return (np.array(starts), np.array(kinds), np.array(idx), np.array(par), sys_mol, atom_starts, pos)


@pytest.mark.parametrize("kind", list(range(6)))
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_check_stage_matches_oracle(kind, seed):
    rng = np.random.default_rng(100 * kind + seed)
    starts, kinds, idx, par, sys_mol, atom_starts, pos = random_case(rng)
    got = stereo_check_flat(kind, torch.from_numpy(pos).cuda(), atom_starts, sys_mol, starts, kinds, idx, par).cpu().numpy()
    want = np.zeros(len(sys_mol), dtype=np.uint8)
    for s, m in enumerate(sys_mol):
        p = pos[atom_starts[s]:atom_starts[s + 1]]
        t0, t1 = starts[m], starts[m + 1]
        want[s] = S.system_fails(kind, p, kinds[t0:t1], idx[t0:t1], par[t0:t1])
    assert np.array_equal(got, want), f"kind {kind}: {int((got != want).sum())} of {len(want)} systems differ"
    assert 0 < want.sum() < len(want)   # both outcomes occur, otherwise the case proves nothing


def test_active_mask_and_failed_is_only_ever_set():
    rng = np.random.default_rng(5)
    starts, kinds, idx, par, sys_mol, atom_starts, pos = random_case(rng, n_sys=64)
    dpos = torch.from_numpy(pos).cuda()
    full = stereo_check_flat(S.TETRAHEDRAL, dpos, atom_starts, sys_mol, starts, kinds, idx, par).cpu().numpy()
    active = (np.arange(64) % 2 == 0).astype(np.uint8)
    half = stereo_check_flat(S.TETRAHEDRAL, dpos, atom_starts, sys_mol, starts, kinds, idx, par, active=active).cpu().numpy()
    assert np.array_equal(half, full * active)


def test_hand_made_cases_and_validation():
    tet = np.array([[0.0, 0, 0, 9], [1, 1, 1, 9], [1, -1, -1, 9], [-1, 1, -1, 9], [-1, -1, 1, 9]], dtype=np.float64)
    flat = tet.copy()
    flat[1:4, 2] = 0.0
    flat[0, 2] = 0.0
    pos = torch.from_numpy(np.concatenate([tet, flat])).cuda()
    out = stereo_check_flat(S.TETRAHEDRAL, pos, [0, 5, 10], [0, 0], [0, 1], [S.TETRAHEDRAL], [[0, 1, 2, 3, 4]], [[0.0, 0.0]])
    assert out.cpu().tolist() == [0, 1]
    with pytest.raises(ValueError):
        stereo_check_flat(S.TETRAHEDRAL, pos[:, :3].contiguous(), [0, 5, 10], [0, 0], [0, 1], [0], [[0, 1, 2, 3, 4]], [[0.0, 0.0]])
    with pytest.raises(ValueError):
        stereo_check_flat(9, pos, [0, 5, 10], [0, 0], [0, 1], [0], [[0, 1, 2, 3, 4]], [[0.0, 0.0]])
