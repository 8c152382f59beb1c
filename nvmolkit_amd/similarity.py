"""Bulk fingerprint similarity on the GPU (same API as the reference's nvmolkit/similarity.py).

``crossTanimotoSimilarity`` / ``crossCosineSimilarity`` return an ``AsyncGpuResult`` holding the
N x M float64 matrix; the ``...MemoryConstrained`` variants return a host numpy array and chunk the
computation when the matrix does not fit on the device (reference: src/similarity.cpp:105-254).
"""

from __future__ import annotations

import numpy as np
import torch

from nvmolkit_amd import _native
from nvmolkit_amd.types import AsyncGpuResult


def _as_words(name: str, fp) -> torch.Tensor:
    t = fp.torch() if isinstance(fp, AsyncGpuResult) else fp
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t, device="cuda")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU")
    if t.dtype not in (torch.int32, torch.uint32):
        raise ValueError(f"{name} must be packed 32-bit words (int32/uint32), got {t.dtype}")
    if t.ndim != 2:
        raise ValueError(f"{name} must be 2-D (n_fingerprints, fp_size/32), got shape {tuple(t.shape)}")
    return t if t.is_contiguous() else t.contiguous()


def _pair(one, two):
    a = _as_words("fingerprint_group_one", one)
    b = a if two is None else _as_words("fingerprint_group_two", two)
    if a.shape[1] != b.shape[1]:
        # reference: nvmolkit/DataStructs.cpp:104-109 -> std::invalid_argument -> ValueError
        raise ValueError(f"Fingerprint sizes do not match: {a.shape[1] * 32} vs {b.shape[1] * 32} bits")
    if a.device != b.device:
        raise ValueError("both fingerprint groups must be on the same device")
    return a, b


def _cross_device(fn_name: str, one, two, stream) -> AsyncGpuResult:
    sptr = _native.stream_ptr(stream)
    first = one if isinstance(one, torch.Tensor) else getattr(one, "torch", lambda: None)()
    device = first.device if isinstance(first, torch.Tensor) and first.is_cuda else torch.device("cuda", torch.cuda.current_device())
    with torch.cuda.device(device), _native.on_stream(stream, device):
        a, b = _pair(one, two)  # any .contiguous() copy is made on the stream the kernel runs on
        out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float64, device=a.device)
        rc = getattr(_native.lib(), fn_name)(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], a.shape[1] * 32,
                                             out.data_ptr(), b.shape[0], sptr)
    _native.check(rc, fn_name)
    return AsyncGpuResult(out)


def _cross_host(metric: int, one, two, max_device_bytes: int = -1) -> np.ndarray:
    a, b = _pair(one, two)
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float64)
    with torch.cuda.device(a.device):
        torch.cuda.current_stream().synchronize()  # inputs may still be in flight on the caller's stream
        rc = _native.lib().nvmk_cross_similarity_host_f64(metric, a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0],
                                                          a.shape[1] * 32, out.ctypes.data, int(max_device_bytes))
    _native.check(rc, "nvmk_cross_similarity_host_f64")
    return out


def crossTanimotoSimilarity(fingerprint_group_one, fingerprint_group_two=None, stream=None) -> AsyncGpuResult:
    """N x M Tanimoto similarities; ``fingerprint_group_two=None`` means all-to-all within group one.

    With a 1 x n_bits first group this equals RDKit's ``BulkTanimotoSimilarity``.
    Reference: nvmolkit/similarity.py:34-71.
    """
    return _cross_device("nvmk_cross_tanimoto_f64", fingerprint_group_one, fingerprint_group_two, stream)


def crossCosineSimilarity(fingerprint_group_one, fingerprint_group_two=None, stream=None) -> AsyncGpuResult:
    """N x M cosine similarities (reference: nvmolkit/similarity.py:113-152)."""
    return _cross_device("nvmk_cross_cosine_f64", fingerprint_group_one, fingerprint_group_two, stream)


def crossTanimotoSimilarityMemoryConstrained(fingerprint_group_one, fingerprint_group_two=None,
                                             max_device_memory_bytes: int = -1) -> np.ndarray:
    """Tanimoto matrix computed on the GPU, returned as a host numpy array, chunked if needed.

    ``max_device_memory_bytes`` exposes the reference's ``CrossSimilarityOptions.maxDeviceMemoryBytes``
    (src/similarity.h:29-32); -1 uses the free memory of the device.
    Reference: nvmolkit/similarity.py:74-105.
    """
    return _cross_host(_native.METRIC_TANIMOTO, fingerprint_group_one, fingerprint_group_two, max_device_memory_bytes)


def crossCosineSimilarityMemoryConstrained(fingerprint_group_one, fingerprint_group_two=None,
                                           max_device_memory_bytes: int = -1) -> np.ndarray:
    """Cosine twin of :func:`crossTanimotoSimilarityMemoryConstrained` (nvmolkit/similarity.py:155-185)."""
    return _cross_host(_native.METRIC_COSINE, fingerprint_group_one, fingerprint_group_two, max_device_memory_bytes)
