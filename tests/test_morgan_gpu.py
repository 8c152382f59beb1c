"""GPU parity: Morgan fingerprints from flattened invariants vs the CPU oracle (bit-exact), modelled on
the reference's tests/test_morgan_fingerprint.cpp:115-164,293-355 (radius x fpSize x batch sweeps) with
RDKit replaced by seeded synthetic graphs + the pinned oracle."""

import numpy as np
import pytest
import torch

import oracle
from nvmolkit_amd import _native
from nvmolkit_amd.fingerprints import MorganFingerprintGenerator, unpack_fingerprint
from tests import util
from tests.test_oracle_morgan import ACID_A, ACID_B, DIOL, PENTANE

pytestmark = pytest.mark.gpu


def gpu_fps(flat, stride, radius, fp_bits):
    gen = MorganFingerprintGenerator(radius, fp_bits)
    res = gen.GetFingerprintsFromInvariants(*flat, max_atoms=stride)
    t = res.torch()
    assert t.dtype == torch.int32 and t.device.type == "cuda"
    return res.numpy().view(np.uint32)


@pytest.mark.parametrize("stride", [32, 64, 128, 256])
@pytest.mark.parametrize("radius", [0, 1, 2, 3, 5])
@pytest.mark.parametrize("fp_bits", [128, 1024, 2048, 4096])
def test_random_batches_bit_exact(stride, radius, fp_bits):
    mols = util.random_molecule_batch(64 if stride < 256 else 24, stride, seed=stride * 10 + radius,
                                      symmetric=(radius % 2 == 0))
    flat = util.flatten_molecules(mols, stride)
    want = oracle.morgan_fingerprints(*flat, stride, radius, fp_bits)
    got = gpu_fps(flat, stride, radius, fp_bits)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("stride", [512, 1024])
@pytest.mark.parametrize("radius", [0, 2, 3])
def test_large_molecule_buckets_bit_exact(stride, radius):
    # 256-1023 atoms: the reference computes these on the CPU; here the bitsets move to a global scratch
    mols = util.random_molecule_batch(6, stride, seed=stride + radius, symmetric=(radius == 2), min_atoms=stride // 2)
    flat = util.flatten_molecules(mols, stride)
    assert int(flat[4].max()) >= stride // 2
    want = oracle.morgan_fingerprints(*flat, stride, radius, 2048)
    assert np.array_equal(gpu_fps(flat, stride, radius, 2048), want)


@pytest.mark.parametrize("batch", [1, 5, 2048])
def test_batch_sizes(batch):
    mols = util.random_molecule_batch(batch, 64, seed=batch, symmetric=True)
    flat = util.flatten_molecules(mols, 64)
    assert np.array_equal(gpu_fps(flat, 64, 2, 2048), oracle.morgan_fingerprints(*flat, 64, 2, 2048))


def test_known_molecules_and_bit_layout():
    mols = [PENTANE, ACID_A, ACID_B, DIOL, ([(6, 4, 0, False)], [])]
    flat = util.flatten_molecules(mols, 32)
    got = gpu_fps(flat, 32, 2, 2048)
    assert np.array_equal(got, oracle.morgan_fingerprints(*flat, 32, 2, 2048))
    assert np.array_equal(got[1], got[2])  # atom-order invariance (test_morgan_fingerprint_ref.cpp:57-58)
    # bit j <-> word j // 32, mask 1 << (j % 32) (nvmolkit/tests/test_fingerprints.py:80-100)
    bits = unpack_fingerprint(torch.from_numpy(got.view(np.int32))).numpy()
    for m, mol in enumerate(mols):
        ai, bi, bx, bo, na = util.flatten_molecules([mol], 32)
        codes, _ = oracle.morgan_environments(ai[0], bi[0], bx[0], bo[0], int(na[0]), 2)
        assert sorted(np.flatnonzero(bits[m]).tolist()) == sorted(set((codes % 2048).tolist()))


def test_highly_symmetric_rings_exercise_dedup():
    # cyclohexane-like and larger all-identical rings: every atom has the same environment each round
    mols = []
    for n in (3, 6, 12, 30):
        atoms = [(6, 2, 0, True)] * n
        bonds = [(i, (i + 1) % n, util.BOND_SINGLE) for i in range(n)]
        mols.append((atoms, bonds))
    flat = util.flatten_molecules(mols, 32)
    for radius in range(0, 6):
        assert np.array_equal(gpu_fps(flat, 32, radius, 1024), oracle.morgan_fingerprints(*flat, 32, radius, 1024))


def test_output_index_scatter_and_empty_molecule(native_lib):
    mols = util.random_molecule_batch(10, 64, seed=77)
    mols[3] = ([], [])
    flat = util.flatten_molecules(mols, 64)
    want = oracle.morgan_fingerprints(*flat, 64, 2, 512)
    perm = np.random.default_rng(0).permutation(10).astype(np.int32)
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    d = [dev(flat[0].view(np.int32)), dev(flat[1].view(np.int32)), dev(flat[2]), dev(flat[3]), dev(flat[4]), dev(perm)]
    out = torch.full((10, 16), -1, dtype=torch.int32, device="cuda")
    rc = native_lib.nvmk_morgan_from_invariants(*(t.data_ptr() for t in d), 10, 64, 2, 512, out.data_ptr(), None)
    _native.check(rc)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    assert np.array_equal(got[perm], want)
    assert not got[perm[3]].any()  # empty molecule -> zero row


@pytest.mark.parametrize("fp_bits", [17, 8192])
def test_invalid_fp_size_raises(fp_bits):
    flat = util.flatten_molecules([PENTANE], 32)
    with pytest.raises(ValueError):
        MorganFingerprintGenerator(2, fp_bits).GetFingerprintsFromInvariants(*flat, max_atoms=32)


def test_bad_stream_and_bucket():
    flat = util.flatten_molecules([PENTANE], 32)
    with pytest.raises(TypeError):
        MorganFingerprintGenerator(2, 1024).GetFingerprintsFromInvariants(*flat, max_atoms=32, stream=5)
    with pytest.raises(ValueError):
        MorganFingerprintGenerator(2, 1024).GetFingerprintsFromInvariants(*flat, max_atoms=48)
