"""Argument checks that are answered before anything touches a GPU (the reference's Python tests expect them "before reaching
the GPU": nvmolkit/tests/test_clustering.py:133-146, test_similarity.py:374-389): wrong stream types, invalid neighbour-list
sizes, invalid metrics and ranges.  CPU tensors throughout."""

import pytest
import torch

from nvmolkit_amd.clustering import butina, fused_butina
from nvmolkit_amd.similarity import crossCosineSimilarity, crossTanimotoSimilarity


@pytest.mark.parametrize("invalid_size", [0, 1, 7, 9, 15, 33, 48, 100, 256])
def test_butina_refuses_invalid_neighborlist_sizes(invalid_size):
    dists = torch.zeros(10, 10, dtype=torch.float64)
    with pytest.raises(ValueError, match="neighborlist_max_size must be one of"):
        butina(dists, 0.1, neighborlist_max_size=invalid_size)


def test_stream_arguments_must_be_torch_streams():
    dists = torch.zeros(10, 10, dtype=torch.float64)
    fps = torch.zeros((4, 64), dtype=torch.int32)
    with pytest.raises(TypeError):
        butina(dists, 0.1, stream=42)
    with pytest.raises(TypeError):
        crossTanimotoSimilarity(fps, stream=42)
    with pytest.raises(TypeError):
        crossCosineSimilarity(fps, stream="0")


def test_butina_needs_a_square_gpu_matrix():
    with pytest.raises((ValueError, RuntimeError, AssertionError)):
        butina(torch.zeros(10, 10, dtype=torch.float64), 0.1)  # a CPU tensor


def test_fused_butina_checks_its_matrix_first():
    with pytest.raises(TypeError):
        fused_butina([[1, 2]], 0.3)
    with pytest.raises(ValueError, match="must be a CUDA tensor"):
        fused_butina(torch.zeros((4, 64), dtype=torch.int32), 0.3)


def test_pack_and_unpack_fingerprints_like_the_reference():
    """nvmolkit/tests/test_fingerprints.py:24-55 on CPU tensors: round trip, a width that is no multiple of 32, the dtype check;
    and the packed layout itself (bit j of a fingerprint = bit j % 32 of word j // 32)."""
    from nvmolkit_amd.fingerprints import pack_fingerprint, unpack_fingerprint

    g = torch.Generator().manual_seed(0)
    fp = torch.randint(0, 2, (10, 128), dtype=torch.bool, generator=g)
    packed = pack_fingerprint(fp)
    assert packed.shape == (10, 4) and packed.dtype == torch.int32
    assert torch.equal(unpack_fingerprint(packed), fp)
    fp = torch.randint(0, 2, (10, 127), dtype=torch.bool, generator=g)
    packed = pack_fingerprint(fp)
    assert packed.shape == (10, 4)
    unpacked = unpack_fingerprint(packed)
    assert unpacked.shape == (10, 128) and torch.equal(unpacked[:, :127], fp) and not unpacked[:, 127].any()
    with pytest.raises(ValueError):
        unpack_fingerprint(torch.randint(0, 2, (10, 32), dtype=torch.int64))
    one = torch.zeros((1, 64), dtype=torch.bool)
    one[0, 33] = True
    assert pack_fingerprint(one).tolist() == [[0, 2]]
    one[0, 31] = True
    assert pack_fingerprint(one).tolist() == [[-(1 << 31), 2]]  # bit 31 is the sign bit of the int32 word


@pytest.mark.parametrize("fp_size", [17, 8192])
def test_morgan_generator_refuses_unsupported_sizes(fp_size):
    """nvmolkit/tests/test_fingerprints.py:57-61 — answered without a molecule or a GPU."""
    from nvmolkit_amd.fingerprints import MorganFingerprintGenerator

    with pytest.raises(Exception):
        MorganFingerprintGenerator(radius=3, fpSize=fp_size).GetFingerprints(["CCO"])
