"""Result / option types of the public API (mirrors the reference's nvmolkit/types.py)."""

from __future__ import annotations

from enum import Enum
from typing import Any, Iterable, List, NamedTuple, Optional

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


class Dense3DResult(NamedTuple):
    """Padded dense view of a :class:`Device3DResult` (reference: nvmolkit/types.py:180-194).

    ``values`` (n_mols, max_confs, max_atoms, 3) float64 with ``pad_value`` in the padding, ``conf_mask``
    (n_mols, max_confs) and ``atom_mask`` (n_mols, max_confs, max_atoms), True where the data is real."""

    values: torch.Tensor
    conf_mask: torch.Tensor
    atom_mask: torch.Tensor


class Device3DResult:
    """Flat, GPU-resident conformer coordinates (+ energies / convergence after a minimisation).

    Same fields and meaning as the reference's ``Device3DResult`` (nvmolkit/types.py:197-246) over
    ``DeviceCoordResult`` (src/conformer/device_coord_result.h:58-67): conformer ``i`` owns rows
    ``values[atom_starts[i]:atom_starts[i+1]]`` of the (total_atoms, 3) float64 ``values``; ``mol_indices[i]`` is its
    input molecule and ``conf_indices[i]`` its index within that molecule; ``n_mols`` counts the input molecules,
    including those without conformers.  All buffers live on GPU ``gpu_id`` and are wrapped as
    :class:`AsyncGpuResult`; synchronise before reading them on the host.
    """

    def __init__(self, values, atom_starts, mol_indices, conf_indices, gpu_id: int, n_mols: int, energies=None,
                 converged=None) -> None:
        wrap = lambda x: x if (x is None or isinstance(x, AsyncGpuResult)) else AsyncGpuResult(x)  # noqa: E731
        self.values = wrap(values)
        self.atom_starts = wrap(atom_starts)
        self.mol_indices = wrap(mol_indices)
        self.conf_indices = wrap(conf_indices)
        self.energies = wrap(energies)
        self.converged = wrap(converged)
        self.gpu_id = int(gpu_id)
        self.n_mols = int(n_mols)
        n_conf = self.mol_indices.torch().numel()
        if self.atom_starts.torch().numel() != n_conf + 1 or self.conf_indices.torch().numel() != n_conf:
            raise ValueError("atom_starts / mol_indices / conf_indices sizes are inconsistent")

    @property
    def num_conformers(self) -> int:
        return int(self.atom_starts.torch().numel()) - 1

    def per_molecule(self) -> List[List[torch.Tensor]]:
        """``result[m][k]``: (n_atoms, 3) view (no copy) of the k-th conformer of input molecule m; molecules without
        conformers get an empty list.  Reading the index tensors synchronises."""
        values = self.values.torch()
        bounds = self.atom_starts.torch().tolist()
        out: List[List[torch.Tensor]] = [[] for _ in range(self.n_mols)]
        for c, m in enumerate(self.mol_indices.torch().tolist()):
            out[m].append(values[bounds[c]:bounds[c + 1]])
        return out

    def dense(self, pad_value: float = float("nan")) -> Dense3DResult:
        """Scatter into a padded (n_mols, max_confs, max_atoms, 3) tensor; masks mark the real entries."""
        values = self.values.torch()
        dev = values.device
        starts = self.atom_starts.torch().to(torch.int64)
        mols = self.mol_indices.torch().to(torch.int64)
        confs = self.conf_indices.torch().to(torch.int64)
        if mols.numel() == 0:
            return Dense3DResult(torch.full((self.n_mols, 0, 0, 3), pad_value, dtype=values.dtype, device=dev),
                                 torch.zeros((self.n_mols, 0), dtype=torch.bool, device=dev),
                                 torch.zeros((self.n_mols, 0, 0), dtype=torch.bool, device=dev))
        n_atoms = starts[1:] - starts[:-1]
        max_confs = int(confs.max().item()) + 1
        max_atoms = int(n_atoms.max().item())
        # flat slot of every atom row in the (n_mols * max_confs * max_atoms) grid
        conf_of_row = torch.repeat_interleave(torch.arange(mols.numel(), device=dev), n_atoms)
        atom_of_row = torch.arange(values.shape[0], device=dev) - starts[conf_of_row]
        slot = (mols[conf_of_row] * max_confs + confs[conf_of_row]) * max_atoms + atom_of_row
        dense = torch.full((self.n_mols * max_confs * max_atoms, 3), pad_value, dtype=values.dtype, device=dev)
        dense[slot] = values
        atom_mask = torch.zeros(self.n_mols * max_confs * max_atoms, dtype=torch.bool, device=dev)
        atom_mask[slot] = True
        conf_mask = torch.zeros(self.n_mols * max_confs, dtype=torch.bool, device=dev)
        conf_mask[mols * max_confs + confs] = True
        return Dense3DResult(dense.view(self.n_mols, max_confs, max_atoms, 3), conf_mask.view(self.n_mols, max_confs),
                             atom_mask.view(self.n_mols, max_confs, max_atoms))
