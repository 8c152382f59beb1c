"""Flattened-term force-field batches: energy, gradient and BFGS minimisation on the GPU.

This is the seam SURVEY.md F7 identifies: the reference flattens RDKit molecules into SoA term arrays
(rdkit_extensions/{mmff,dist_geom}_flattened_builder.cpp) and every kernel works on those arrays.  The classes
here take exactly such arrays (numpy), keep them resident on the device and drive the C ABI
(nvmk_ff_energy / nvmk_ff_gradient / nvmk_bfgs_minimize).  Counterpart of the reference's BatchedForcefield
interface (src/forcefields/batched_forcefield.h:74-149) and BfgsBatchMinimizer (src/minimizer/bfgs_minimize.h).
"""

from __future__ import annotations

import ctypes
from typing import Sequence

import numpy as np
import torch

from nvmolkit_amd import _native

DG, ETK, MMFF, QUARTIC = _native.FF_DG, _native.FF_ETK, _native.FF_MMFF, _native.FF_QUARTIC
#: (n_idx, n_par) of every term group, per kind (see include/nvmolkit_amd.h)
GROUP_LAYOUT = {
    DG: [(2, 3), (4, 2), (1, 0)],
    ETK: [(4, 12), (4, 4), (2, 4), (2, 4), (3, 2), (2, 4)],
    MMFF: [(2, 2), (3, 3), (3, 5), (4, 1), (4, 3), (2, 2), (2, 3)],
    QUARTIC: [],
}
DIM = {DG: 4, ETK: 4, MMFF: 3, QUARTIC: 4}


class FlatForcefieldBatch:
    """`n_systems` independent systems with their term tables resident on one GPU.

    Args:
        kind: DG, ETK, MMFF or QUARTIC.
        atom_starts: (n_systems + 1,) CSR offsets of each system's atoms.
        groups: one ``(starts, idx, par)`` triple per term group of the kind — ``starts`` (n_systems + 1,),
            ``idx`` (n_terms, n_idx) LOCAL atom indices, ``par`` (n_terms, n_par) float64.
    """

    def __init__(self, kind: int, atom_starts, groups: Sequence[tuple], device="cuda"):
        if kind not in GROUP_LAYOUT:
            raise ValueError(f"unknown force-field kind {kind}")
        layout = GROUP_LAYOUT[kind]
        if len(groups) != len(layout):
            raise ValueError(f"kind {kind} needs {len(layout)} term groups, got {len(groups)}")
        self.kind = kind
        self.dim = DIM[kind]
        self.device = torch.device(device)
        self.atom_starts_host = np.ascontiguousarray(atom_starts, dtype=np.int32)
        self.n_systems = len(self.atom_starts_host) - 1
        if self.n_systems < 0 or np.any(np.diff(self.atom_starts_host) < 0):
            raise ValueError("atom_starts must be a non-decreasing CSR offset array")
        self._keep = [torch.from_numpy(self.atom_starts_host).to(self.device)]
        self._c = _native.FFBatch()
        self._c.kind = kind
        self._c.n_systems = self.n_systems
        self._c.atom_starts = self._keep[0].data_ptr()
        for g, ((n_idx, n_par), (starts, idx, par)) in enumerate(zip(layout, groups)):
            starts = np.ascontiguousarray(starts, dtype=np.int32)
            idx = np.ascontiguousarray(idx, dtype=np.int32).reshape(-1, n_idx)
            par = (np.ascontiguousarray(par, dtype=np.float64).reshape(-1, n_par) if n_par else np.zeros((len(idx), 0)))
            if len(starts) != self.n_systems + 1 or (len(starts) and starts[-1] != len(idx)) or len(par) != len(idx):
                raise ValueError(f"term group {g}: inconsistent starts / idx / par sizes")
            t = [torch.from_numpy(starts).to(self.device), torch.from_numpy(idx.copy()).to(self.device),
                 torch.from_numpy(np.ascontiguousarray(par)).to(self.device)]
            self._keep.extend(t)
            self._c.groups[g].starts = t[0].data_ptr()
            self._c.groups[g].idx = t[1].data_ptr() if len(idx) else None
            self._c.groups[g].par = t[2].data_ptr() if par.size else None

    @property
    def n_atoms_total(self) -> int:
        return int(self.atom_starts_host[-1]) if self.n_systems >= 0 and len(self.atom_starts_host) else 0

    def _check_pos(self, pos: torch.Tensor) -> torch.Tensor:
        if not isinstance(pos, torch.Tensor) or not pos.is_cuda or pos.dtype != torch.float64:
            raise ValueError("positions must be a float64 CUDA tensor")
        if pos.numel() != self.n_atoms_total * self.dim:
            raise ValueError(f"positions must hold {self.n_atoms_total} atoms x {self.dim} coordinates")
        if not pos.is_contiguous():
            raise ValueError("positions must be contiguous")
        return pos

    @staticmethod
    def _mask(active):
        return None if active is None else active.to(torch.uint8).contiguous()

    def compute_energy(self, pos: torch.Tensor, w0: float = 1.0, w1: float = 1.0, active=None, stream=None) -> torch.Tensor:
        """Per-system energies (reference: BatchedForcefield::computeEnergy)."""
        self._check_pos(pos)
        out = torch.zeros(max(self.n_systems, 0), dtype=torch.float64, device=self.device)
        m = self._mask(active)
        rc = _native.lib().nvmk_ff_energy(ctypes.byref(self._c), float(w0), float(w1), pos.data_ptr(),
                                          m.data_ptr() if m is not None else None, out.data_ptr(),
                                          _native.stream_ptr(stream))
        _native.check(rc, "nvmk_ff_energy")
        return out

    def compute_gradient(self, pos: torch.Tensor, w0: float = 1.0, w1: float = 1.0, active=None, stream=None) -> torch.Tensor:
        """Gradient with the layout of ``pos`` (reference: BatchedForcefield::computeGradients)."""
        self._check_pos(pos)
        grad = torch.zeros_like(pos)
        m = self._mask(active)
        rc = _native.lib().nvmk_ff_gradient(ctypes.byref(self._c), float(w0), float(w1), pos.data_ptr(),
                                            m.data_ptr() if m is not None else None, grad.data_ptr(),
                                            _native.stream_ptr(stream))
        _native.check(rc, "nvmk_ff_gradient")
        return grad

    def minimize(self, pos: torch.Tensor, max_iters: int = 200, grad_tol: float = 1e-4, scale_grads: bool = True,
                 w0: float = 1.0, w1: float = 1.0, active=None, stream=None):
        """BFGS-minimise every (active) system in place.

        Returns ``(energies, statuses, iterations)``; status 0 = converged (reference: BfgsBatchMinimizer::minimize,
        src/minimizer/bfgs_minimize.cu:978-1084; fused kernel bfgs_minimize_permol_kernels.cu:426-745)."""
        self._check_pos(pos)
        n = max(self.n_systems, 0)
        energies = torch.zeros(n, dtype=torch.float64, device=self.device)
        statuses = torch.full((n,), -1, dtype=torch.int16, device=self.device)
        iters = torch.zeros(n, dtype=torch.int32, device=self.device)
        m = self._mask(active)
        rc = _native.lib().nvmk_bfgs_minimize(ctypes.byref(self._c), self.atom_starts_host.ctypes.data, float(w0), float(w1),
                                              int(max_iters), float(grad_tol), int(bool(scale_grads)), pos.data_ptr(),
                                              m.data_ptr() if m is not None else None, energies.data_ptr(),
                                              statuses.data_ptr(), iters.data_ptr(), _native.stream_ptr(stream))
        _native.check(rc, "nvmk_bfgs_minimize")
        return energies, statuses, iters
