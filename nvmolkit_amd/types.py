"""Result / option types of the public API (mirrors the reference's nvmolkit/types.py)."""

from __future__ import annotations

from enum import Enum
from typing import Any, Iterable, List, Optional

import torch


class AsyncGpuResult:
    """Handle to a GPU-resident result that may still be being computed on its stream.

    Same contract as the reference (nvmolkit/types.py:125-162): exposes
    ``__cuda_array_interface__``, ``.torch()`` (asynchronous view) and ``.numpy()`` (blocking copy).
    Ownership differs by design: the reference wraps a C++ ``PyArray`` that frees with
    ``cudaFreeAsync``; here the buffer is a torch allocation, so torch's caching allocator owns it.
    """

    def __init__(self, obj, gpu_id: Optional[int] = None):
        if isinstance(obj, AsyncGpuResult):
            obj = obj.arr
        if isinstance(obj, torch.Tensor):
            self.arr = obj
        else:
            if not hasattr(obj, "__cuda_array_interface__"):
                raise TypeError(f"Object {obj} does not have a __cuda_array_interface__ attribute")
            device = "cuda" if gpu_id is None else f"cuda:{int(gpu_id)}"
            self.arr = torch.as_tensor(obj, device=device)

    @property
    def __cuda_array_interface__(self):
        return self.arr.__cuda_array_interface__

    @property
    def device(self):
        return self.arr.device

    def torch(self) -> torch.Tensor:
        """The underlying tensor; consuming it on another stream needs a sync."""
        return self.arr

    def numpy(self):
        """Blocking copy to a numpy array."""
        if self.arr.is_cuda:
            torch.cuda.synchronize(self.arr.device)
        return self.arr.cpu().numpy()


class HardwareOptions:
    """Threading / batching / device selection for the molecule-batch APIs.

    Field-for-field the reference's ``HardwareOptions`` (nvmolkit/types.py:26-122) over
    ``BatchHardwareOptions`` (src/hardware_options.h:26-35): -1 / empty mean "choose automatically".
    """

    _FIELDS = ("preprocessingThreads", "batchSize", "batchesPerGpu", "gpuIds")

    def __init__(self, preprocessingThreads: int = -1, batchSize: int = -1, batchesPerGpu: int = -1,
                 gpuIds: Iterable[int] | None = None) -> None:
        self.preprocessingThreads = preprocessingThreads
        self.batchSize = batchSize
        self.batchesPerGpu = batchesPerGpu
        self.gpuIds = gpuIds if gpuIds is not None else []

    @property
    def preprocessingThreads(self) -> int:
        return self._preprocessingThreads

    @preprocessingThreads.setter
    def preprocessingThreads(self, value: int) -> None:
        self._preprocessingThreads = int(value)

    @property
    def batchSize(self) -> int:
        return self._batchSize

    @batchSize.setter
    def batchSize(self, value: int) -> None:
        self._batchSize = int(value)

    @property
    def batchesPerGpu(self) -> int:
        return self._batchesPerGpu

    @batchesPerGpu.setter
    def batchesPerGpu(self, value: int) -> None:
        value = int(value)
        if value != -1 and value <= 0:
            raise ValueError("batchesPerGpu must be greater than 0 or -1 for automatic")
        self._batchesPerGpu = value

    @property
    def gpuIds(self) -> List[int]:
        return list(self._gpuIds)

    @gpuIds.setter
    def gpuIds(self, value: Iterable[int]) -> None:
        self._gpuIds = [int(v) for v in value]

    def to_dict(self) -> dict[str, Any]:
        return {name: getattr(self, name) for name in self._FIELDS}

    @classmethod
    def from_dict(cls, data: dict[str, Any]) -> "HardwareOptions":
        unknown = set(data) - set(cls._FIELDS)
        if unknown:
            raise KeyError(f"Unknown HardwareOptions keys: {sorted(unknown)}")
        return cls(**{k: data[k] for k in cls._FIELDS if k in data})

    def __repr__(self) -> str:
        return f"HardwareOptions({', '.join(f'{k}={getattr(self, k)!r}' for k in self._FIELDS)})"


class CoordinateOutput(Enum):
    """Where conformer-producing APIs leave coordinates (reference: nvmolkit/types.py:165-178)."""

    RDKIT_CONFORMERS = "rdkit"
    DEVICE = "device"
