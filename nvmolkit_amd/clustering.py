"""Butina clustering on the GPU (same API as the reference's nvmolkit/clustering.py).

``butina`` clusters a dense distance matrix; ``fused_butina`` clusters packed fingerprints without
ever materialising the N x N matrix (the only way to reach 1M points).
"""

from __future__ import annotations

import ctypes

import numpy as np
import torch

from nvmolkit_amd import _native
from nvmolkit_amd.types import AsyncGpuResult

_VALID_NEIGHBORLIST_SIZES = frozenset({8, 16, 24, 32, 64, 128})
_METRICS = {"tanimoto": _native.METRIC_TANIMOTO, "cosine": _native.METRIC_COSINE}


def butina(distance_matrix, cutoff: float, neighborlist_max_size: int = 64, return_centroids: bool = False,
           stream=None):
    """Taylor-Butina clustering of a square distance matrix (reference: nvmolkit/clustering.py:41-96).

    Points are neighbours when ``distance <= cutoff`` (reference kernel: src/butina.cu:1043-1051).
    Returns an ``AsyncGpuResult`` of int32 cluster ids (id 0 = largest cluster), plus an
    ``AsyncGpuResult`` of centroid indices per cluster id when ``return_centroids`` is set.
    ``neighborlist_max_size`` is validated like the reference; it only tunes the reference's
    small-cluster phase and does not change results here.
    """
    if neighborlist_max_size not in _VALID_NEIGHBORLIST_SIZES:
        raise ValueError(
            f"neighborlist_max_size must be one of {sorted(_VALID_NEIGHBORLIST_SIZES)}, got {neighborlist_max_size}")
    sptr = _native.stream_ptr(stream)
    d = distance_matrix.torch() if isinstance(distance_matrix, AsyncGpuResult) else distance_matrix
    if not isinstance(d, torch.Tensor) or not d.is_cuda:
        raise ValueError("distance_matrix must be a GPU tensor or AsyncGpuResult")
    if d.ndim != 2 or d.shape[0] != d.shape[1]:
        raise ValueError(f"distance_matrix must be square, got shape {tuple(d.shape)}")
    n = d.shape[0]
    with torch.cuda.device(d.device), _native.on_stream(stream, d.device):  # conversions and outputs on the kernel's stream
        if d.dtype != torch.float64:
            d = d.to(torch.float64)
        d = d.contiguous()
        clusters = torch.empty(n, dtype=torch.int32, device=d.device)
        centroids = torch.empty(n, dtype=torch.int32, device=d.device)
        n_clusters = ctypes.c_int64(0)
        rc = _native.lib().nvmk_butina_dense(d.data_ptr(), None, n, float(cutoff), int(neighborlist_max_size),
                                             clusters.data_ptr(), centroids.data_ptr(), ctypes.byref(n_clusters), sptr)
    _native.check(rc, "nvmk_butina_dense")
    if return_centroids:
        return AsyncGpuResult(clusters), AsyncGpuResult(centroids[:n_clusters.value])
    return AsyncGpuResult(clusters)


def _check_fingerprint_matrix(name: str, x) -> None:
    # reference: nvmolkit/_fusedButina.py:43-52
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not x.is_cuda:
        raise ValueError(f"{name} must be a CUDA tensor")
    if x.dtype != torch.int32:
        raise ValueError(f"{name} must have dtype int32")
    if x.ndim != 2:
        raise ValueError(f"{name} must be 2D, got shape={tuple(x.shape)}")


def update_neighbor_counts(x: torch.Tensor, y: torch.Tensor, neighbors: torch.Tensor, threshold: float,
                           subtract: bool = False, metric: str = "tanimoto") -> None:
    """neighbors[i] += (or -=) #{j : sim(x_i, y_j) >= threshold} (reference: nvmolkit/_fusedButina.py:249-289)."""
    _check_fingerprint_matrix("x", x)
    _check_fingerprint_matrix("y", y)
    if neighbors.dtype != torch.int32 or neighbors.ndim != 1 or neighbors.numel() != x.shape[0]:
        raise ValueError(f"neighbors must be a 1D int32 tensor of length {x.shape[0]}")
    if x.device != y.device or x.device != neighbors.device:
        raise ValueError("x, y, and neighbors must be on the same CUDA device")
    if x.shape[1] != y.shape[1]:
        raise ValueError("x and y must have the same feature dimension")
    if metric not in _METRICS:
        raise ValueError(f"metric must be one of ['tanimoto', 'cosine'], got {metric}")
    x = x.contiguous()
    y = y.contiguous()
    with torch.cuda.device(x.device):
        rc = _native.lib().nvmk_neighbor_counts(_METRICS[metric], x.data_ptr(), None, x.shape[0], y.data_ptr(), None,
                                                y.shape[0], x.shape[1] * 32, float(threshold), -1 if subtract else 1,
                                                neighbors.data_ptr(), _native.stream_ptr(None))
    _native.check(rc, "nvmk_neighbor_counts")


def fused_butina(x: torch.Tensor, cutoff: float, return_centroids: bool = False, stream=None,
                 metric: str = "tanimoto"):
    """Matrix-free Butina clustering of packed fingerprints (reference: nvmolkit/clustering.py:99-189).

    Returns ``(clusters, cluster_sizes)`` — a list of tuples (centroid first) and the cumulative size
    list starting at 0 — plus the centroid list when ``return_centroids`` is set.  Neighbours are
    rows with ``float32(similarity) >= float32(1 - cutoff)``.
    """
    _check_fingerprint_matrix("x", x)
    if metric not in _METRICS:
        raise ValueError(f"metric must be one of ['tanimoto', 'cosine'], got {metric}")
    sptr = _native.stream_ptr(stream)
    if cutoff < 0 or cutoff > 1:
        raise ValueError(f"cutoff must be in [0, 1], got {cutoff}")
    n = x.shape[0]
    idx = np.empty(max(n, 1), dtype=np.int32)
    offs = np.zeros(n + 1, dtype=np.int64)
    cent = np.empty(max(n, 1), dtype=np.int32)
    n_clusters = ctypes.c_int64(0)
    with torch.cuda.device(x.device), _native.on_stream(stream, x.device):
        x = x.contiguous()
        rc = _native.lib().nvmk_butina_fused(_METRICS[metric], x.data_ptr(), n, x.shape[1] * 32, float(cutoff),
                                             idx.ctypes.data, offs.ctypes.data, cent.ctypes.data,
                                             ctypes.byref(n_clusters), sptr)
    _native.check(rc, "nvmk_butina_fused")
    k = n_clusters.value
    bounds = offs[:k + 1].tolist()
    flat = memoryview(idx)  # tuple() of a slice makes the Python ints directly (no 1M-element list in between: -20 % here)
    clusters = [tuple(flat[bounds[i]:bounds[i + 1]]) for i in range(k)]
    if return_centroids:
        return clusters, bounds, cent[:k].tolist()
    return clusters, bounds
