"""GPU parity of the conformer RMSD matrices and RMS pruning with the oracle (tolerance 1e-9 A: same fp64 sums in another
order, closed-form eigenvalues vs LAPACK), plus the reference's API behaviour (nvmolkit/tests/test_conformer_rmsd.py)."""

import numpy as np
import pytest
import torch

from oracle import rmsd as orc
from nvmolkit_amd.conformerRmsd import (GetConformerRMSMatrix, GetConformerRMSMatrixBatch, conformer_rms_matrix_flat,
                                        prune_conformers)
from nvmolkit_amd.types import AsyncGpuResult, Device3DResult
from tests.test_device_chain_gpu import FakeConformer, FakeMol

pytestmark = pytest.mark.gpu


def confs(rng, n_confs, n_atoms, spread=1.0):
    base = rng.normal(size=(n_atoms, 3)) * 2.0
    return np.stack([base + spread * rng.normal(size=base.shape) for _ in range(n_confs)])


@pytest.mark.parametrize("prealigned", [False, True])
@pytest.mark.parametrize("shape", [(2, 1), (2, 2), (3, 3), (7, 20), (20, 64), (5, 65), (12, 200)])
def test_matrix_matches_oracle(shape, prealigned):
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    c = confs(rng, *shape)
    got = conformer_rms_matrix_flat([torch.from_numpy(c).cuda()], prealigned)[0].cpu().numpy()
    # rank-deficient cross-covariances (2 or 3 atoms) square-root a cancellation error: 1e-7 there, 1e-9 otherwise
    np.testing.assert_allclose(got, orc.rms_matrix(c, prealigned), rtol=0, atol=1e-7 if shape[1] <= 3 else 1e-9)


def test_batch_is_one_launch_with_ragged_molecules():
    rng = np.random.default_rng(9)
    batch = [confs(rng, 4, 10), np.zeros((0, 7, 3)), confs(rng, 1, 5), confs(rng, 9, 33), confs(rng, 2, 3)]
    got = conformer_rms_matrix_flat([torch.from_numpy(c).cuda() for c in batch])
    assert [g.numel() for g in got] == [6, 0, 0, 36, 1]
    for g, c in zip(got, batch):
        np.testing.assert_allclose(g.cpu().numpy(), orc.rms_matrix(c), rtol=0, atol=1e-9)


def test_rigid_motions_and_mirror_images():
    rng = np.random.default_rng(4)
    a = rng.normal(size=(30, 3))
    q, r = np.linalg.qr(rng.normal(size=(3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    c = np.stack([a, a @ q.T + 3.0, a * np.array([1.0, 1.0, -1.0])])
    m = conformer_rms_matrix_flat([torch.from_numpy(c).cuda()])[0].cpu().numpy()
    assert m[0] < 1e-6                       # (1, 0): rotated + translated copy
    assert m[1] > 0.1 and m[2] > 0.1         # mirror image vs both
    assert m[1] == pytest.approx(m[2], abs=1e-6)


def test_reference_api_on_duck_typed_molecules():
    rng = np.random.default_rng(6)
    c = confs(rng, 5, 12)
    mol = FakeMol([FakeConformer(k, c[k]) for k in range(5)])
    res = GetConformerRMSMatrix(mol)
    assert isinstance(res, AsyncGpuResult) and res.torch().shape == (10,)
    np.testing.assert_allclose(res.numpy(), orc.rms_matrix(c), rtol=0, atol=1e-9)
    np.testing.assert_allclose(GetConformerRMSMatrix(mol, prealigned=True).numpy(), orc.rms_matrix(c, True), rtol=0, atol=1e-9)
    single = FakeMol([FakeConformer(0, c[0])])
    out = GetConformerRMSMatrixBatch([mol, single])
    assert [o.torch().numel() for o in out] == [10, 0]
    with pytest.raises(ValueError):
        GetConformerRMSMatrix(None)
    with pytest.raises(ValueError):
        GetConformerRMSMatrixBatch([mol, None])
    with pytest.raises(TypeError):
        GetConformerRMSMatrix(mol, stream=3)


def test_pruning_matches_oracle_and_compacts_the_result():
    rng = np.random.default_rng(12)
    sizes = [8, 5, 11]
    n_confs = [6, 0, 9]
    per_mol = []
    for n, k in zip(sizes, n_confs):
        base = rng.normal(size=(max(k, 1), n, 3)) * 1.5
        c = np.stack([base[rng.integers(0, max(1, k // 2))] + 0.02 * rng.normal(size=(n, 3)) for _ in range(k)]) if k else np.zeros((0, n, 3))
        per_mol.append(c)
    values = torch.from_numpy(np.concatenate([c.reshape(-1, 3) for c in per_mol])).cuda()
    starts = np.concatenate([[0], np.cumsum([n for n, k in zip(sizes, n_confs) for _ in range(k)])]).astype(np.int32)
    mols = np.repeat(np.arange(3), n_confs).astype(np.int32)
    cidx = np.concatenate([np.arange(k) for k in n_confs]).astype(np.int32)
    energies = torch.arange(len(mols), dtype=torch.float64, device="cuda")
    dev = Device3DResult(values, torch.from_numpy(starts).cuda(), torch.from_numpy(mols).cuda(), torch.from_numpy(cidx).cuda(),
                         0, 3, energies=energies, converged=torch.ones(len(mols), dtype=torch.int8, device="cuda"))
    thr = 0.3
    pruned = prune_conformers(dev, thr)
    want = np.concatenate([orc.prune(c, thr) for c in per_mol if len(c)])
    assert 0 < want.sum() < len(want)
    assert pruned.num_conformers == int(want.sum()) and pruned.n_mols == 3
    assert pruned.mol_indices.torch().tolist() == mols[want].tolist()
    per = pruned.per_molecule()
    kept_idx = np.nonzero(want)[0]
    for slot, src in enumerate(kept_idx):
        m = int(mols[src])
        k = pruned.conf_indices.torch()[slot].item()
        assert torch.equal(per[m][k].cpu(), values[starts[src]:starts[src + 1]].cpu())
    assert pruned.energies.torch().tolist() == kept_idx.astype(float).tolist()
    assert [len(p) for p in per] == [int(orc.prune(c, thr).sum()) if len(c) else 0 for c in per_mol]
    assert prune_conformers(dev, 0.0) is dev                     # pruning disabled
    heavy = [np.arange(0, n, 2) for n in sizes]                  # an atom subset changes the RMSD it prunes on
    sub = prune_conformers(dev, thr, atom_subsets=heavy)
    want_sub = np.concatenate([orc.prune(c[:, h], thr) for c, h in zip(per_mol, heavy) if len(c)])
    assert sub.num_conformers == int(want_sub.sum())


def test_embedding_with_rms_pruning():
    """ETKDG -> RMS pruning on the device (EmbedParameters.pruneRmsThresh): the survivors of every molecule are mutually
    farther apart than the threshold, and a huge threshold leaves exactly one conformer per molecule."""
    from nvmolkit_amd.embedMolecules import FlatMolecule, FlatMoleculeSet, embed_flat
    from nvmolkit_amd.types import CoordinateOutput
    from tests import util

    rng = np.random.default_rng(8)
    molset = FlatMoleculeSet([FlatMolecule(**util.synthetic_embed_molecule(rng, n, True)[0]) for n in (6, 9, 12)])
    kw = dict(confs_per_molecule=6, max_iterations=20, enforce_chirality=False, seed=2, output=CoordinateOutput.DEVICE)
    full = embed_flat(molset, **kw)
    pruned = embed_flat(molset, prune_rms_thresh=0.5, **kw)
    one = embed_flat(molset, prune_rms_thresh=1e6, **kw)
    assert full.num_conformers == 18 and [len(p) for p in one.per_molecule()] == [1, 1, 1]
    assert 3 <= pruned.num_conformers <= 18
    for views in pruned.per_molecule():
        c = np.stack([v.cpu().numpy() for v in views])
        m = orc.rms_matrix(c)
        assert (m >= 0.5 - 1e-9).all()
    with pytest.raises(ValueError):
        embed_flat(molset, confs_per_molecule=2, prune_rms_thresh=0.5)


def permuted_copies(rng, n_confs, n_atoms, matches, spread):
    """Conformers around one shape, each relabelled by a random self match: plain RMSD sees them far apart."""
    base = rng.normal(size=(n_atoms, 3)) * 2.0
    out = []
    for _ in range(n_confs):
        c = base + spread * rng.normal(size=base.shape)
        full = np.arange(n_atoms)
        full[matches[0]] = matches[rng.integers(len(matches))]
        out.append(c[full])
    return np.stack(out)


def random_matches(rng, n_atoms, length, k):
    first = rng.permutation(n_atoms)[:length]
    return np.stack([first] + [rng.permutation(first) for _ in range(k - 1)]).astype(np.int32)


@pytest.mark.parametrize("shape", [(2, 2, 2, 1), (4, 9, 9, 2), (7, 20, 14, 12), (6, 70, 64, 5), (5, 130, 65, 3), (3, 40, 40, 144)])
def test_symmetric_matrix_matches_oracle(shape):
    n_confs, n_atoms, length, k = shape
    rng = np.random.default_rng(sum(shape))
    matches = random_matches(rng, n_atoms, length, k)
    c = permuted_copies(rng, n_confs, n_atoms, matches, 0.3)
    from nvmolkit_amd.conformerRmsd import conformer_rms_matrix_sym_flat

    got = conformer_rms_matrix_sym_flat([torch.from_numpy(c).cuda()], [matches])[0].cpu().numpy()
    np.testing.assert_allclose(got, orc.rms_matrix_sym(c, matches), rtol=0, atol=1e-7 if length <= 3 else 1e-9)
    if k == 1 and length == n_atoms:
        np.testing.assert_allclose(got, orc.rms_matrix(c[:, matches[0]]), rtol=0, atol=1e-7)


def test_symmetric_batch_is_one_launch_with_ragged_molecules_and_match_tables():
    from nvmolkit_amd.conformerRmsd import conformer_rms_matrix_sym_flat

    rng = np.random.default_rng(19)
    sizes = [(4, 10, 10, 3), (0, 7, 5, 2), (1, 5, 5, 1), (9, 33, 20, 24), (2, 3, 3, 6)]
    tables = [random_matches(rng, na, ln, k) for _, na, ln, k in sizes]
    batch = [permuted_copies(rng, nc, na, t, 0.2) if nc else np.zeros((0, na, 3)) for (nc, na, _, _), t in zip(sizes, tables)]
    got = conformer_rms_matrix_sym_flat([torch.from_numpy(c).cuda() for c in batch], tables)
    assert [g.numel() for g in got] == [6, 0, 0, 36, 1]
    for g, c, t in zip(got, batch, tables):
        np.testing.assert_allclose(g.cpu().numpy(), orc.rms_matrix_sym(c, t), rtol=0, atol=1e-7)
    with pytest.raises(ValueError):
        conformer_rms_matrix_sym_flat([torch.from_numpy(batch[0]).cuda()], [np.array([[0, 1, 99]])])
    with pytest.raises(ValueError):
        conformer_rms_matrix_sym_flat([torch.from_numpy(batch[0]).cuda()], [])


def test_pruning_with_symmetry_drops_relabelled_copies():
    """useSymmetryForPruning (reference: getMolSelfMatches + _isConfFarFromRest): conformers that differ by a permutation of
    equivalent atoms are one conformer.  para-xylene from the library's own SMILES ingestion, eight heavy atoms, four self
    matches; hydrogens ride along in the conformers but not in the RMSD."""
    from nvmolkit_amd.fingerprints import SmilesSet

    s = SmilesSet(["Cc1ccc(C)cc1", "CCO"], perceive_aromaticity=True)
    tables = [s.self_matches(0), s.self_matches(1)]
    assert tables[0].shape == (4, 8) and tables[1].shape == (1, 3)
    rng = np.random.default_rng(77)
    per_mol = [permuted_copies(rng, 8, 18, tables[0], 0.01), permuted_copies(rng, 5, 9, tables[1], 0.6)]
    values = torch.from_numpy(np.concatenate([c.reshape(-1, 3) for c in per_mol])).cuda()
    starts = np.concatenate([[0], np.cumsum([c.shape[1] for c in per_mol for _ in range(len(c))])]).astype(np.int32)
    mols = np.concatenate([np.full(len(c), m) for m, c in enumerate(per_mol)]).astype(np.int32)
    cidx = np.concatenate([np.arange(len(c)) for c in per_mol]).astype(np.int32)
    dev = Device3DResult(values, torch.from_numpy(starts).cuda(), torch.from_numpy(mols).cuda(), torch.from_numpy(cidx).cuda(), 0, 2)
    thr = 0.25
    with_sym = prune_conformers(dev, thr, self_matches=tables)
    want = [orc.prune_sym(c, thr, t) for c, t in zip(per_mol, tables)]
    assert [len(p) for p in with_sym.per_molecule()] == [int(w.sum()) for w in want]
    assert want[0].sum() == 1                                    # every relabelled copy is recognised
    heavy_only = prune_conformers(dev, thr, atom_subsets=[t[0] for t in tables])
    assert len(heavy_only.per_molecule()[0]) > 1                  # without the matches the copies look different
    assert len(heavy_only.per_molecule()[1]) == len(with_sym.per_molecule()[1])      # ethanol has no symmetry to use
    with pytest.raises(ValueError):
        prune_conformers(dev, thr, atom_subsets=[None, None], self_matches=tables)
