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
