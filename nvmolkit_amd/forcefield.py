"""Flattened-term force-field batches: energy, gradient and BFGS minimisation on the GPU.

This is the seam SURVEY.md F7 identifies: the reference flattens RDKit molecules into SoA term arrays
(rdkit_extensions/{mmff,dist_geom}_flattened_builder.cpp) and every kernel works on those arrays.  The classes
here take exactly such arrays (numpy), keep them resident on the device and drive the C ABI
(nvmk_ff_energy / nvmk_ff_gradient / nvmk_bfgs_minimize).  Counterpart of the reference's BatchedForcefield
interface (src/forcefields/batched_forcefield.h:74-149) and BfgsBatchMinimizer (src/minimizer/bfgs_minimize.h).
"""

from __future__ import annotations

import ctypes
import os
from typing import Sequence

import numpy as np
import torch

from nvmolkit_amd import _native
from nvmolkit_amd.types import Device3DResult

DG, ETK, MMFF, QUARTIC, UFF = _native.FF_DG, _native.FF_ETK, _native.FF_MMFF, _native.FF_QUARTIC, _native.FF_UFF
#: (n_idx, n_par) of every term group, per kind (see include/nvmolkit_amd.h)
GROUP_LAYOUT = {
    DG: [(2, 3), (4, 2), (1, 0)],
    ETK: [(4, 12), (4, 4), (2, 4), (2, 4), (3, 2), (2, 4)],
    MMFF: [(2, 2), (3, 3), (3, 5), (4, 1), (4, 3), (2, 2), (2, 3)],
    QUARTIC: [],
    UFF: [(2, 2), (3, 6), (4, 3), (4, 4), (2, 3)],
}
DIM = {DG: 4, ETK: 3, MMFF: 3, QUARTIC: 4, UFF: 3}
#: pair-term groups whose rows are re-ordered on the device when a batch is built (see :func:`diagonal_pair_order`);
#: the small 1-2 / 1-3 groups keep the caller's order (ETK groups 2 and 3 are tied to per-system reference distances)
PAIR_ORDER_GROUPS = {DG: (0,), ETK: (5,), MMFF: (5, 6), UFF: (4,), QUARTIC: ()}
# MMFF / UFF batches may append up to four constraint groups (distance, position, angle, torsion; include/nvmolkit_amd.h)
CONSTRAINT_LAYOUT = [(2, 3), (1, 5), (3, 3), (4, 3)]


def pair_order_enabled() -> bool:
    """``NVMK_PAIR_ORDER=input`` keeps the caller's row order of the pair tables (A/B switch for measurements)."""
    return os.environ.get("NVMK_PAIR_ORDER", "diagonal") != "input"


def diagonal_pair_order(starts: torch.Tensor, idx: torch.Tensor, par: torch.Tensor):
    """Re-order the rows of a pair-term group inside every system by (|j - i|, min(i, j)).

    RDKit (and every natural builder) emits the O(N^2) pair lists i-major: 64 consecutive terms share atom i, so the 64
    lanes of a wavefront accumulate their forces into the SAME three LDS words and the hardware serialises the atomic
    adds lane by lane.  Along a diagonal of the pair matrix consecutive terms touch distinct atoms on both sides.  The
    energy and gradient are sums over terms, so only the floating-point summation order changes.  Runs on the device
    (one key computation + one sort per group and batch).
    """
    n_terms = idx.shape[0]
    if n_terms == 0:
        return idx, par
    counts = (starts[1:] - starts[:-1]).to(torch.int64)
    seg = torch.repeat_interleave(torch.arange(counts.numel(), device=idx.device, dtype=torch.int64), counts, output_size=n_terms)
    a, b = idx[:, 0].to(torch.int64), idx[:, 1].to(torch.int64)
    lo = torch.minimum(a, b)
    key = (seg << 40) | ((torch.maximum(a, b) - lo) << 20) | lo
    perm = torch.argsort(key, stable=True)
    return idx[perm].contiguous(), par[perm].contiguous()


def mmff_merge_enabled() -> bool:
    """``NVMK_MMFF_MERGE=0`` keeps van der Waals and electrostatics as separate tables (A/B switch)."""
    return os.environ.get("NVMK_MMFF_MERGE", "1") != "0"


def merge_mmff_nonbonded(vdw, ele):
    """Device-side merge of the MMFF van der Waals group ``(starts, idx, par(R*, eps))`` and electrostatic group
    ``(starts, idx, par(chargeTerm, dielModel, is1_4))`` into one table ``(starts, idx, par(R*, eps, chargeTerm, dielModel,
    is1_4))`` with one row per van der Waals pair.  Returns ``None`` when the lists cannot be merged (an electrostatic pair
    without a van der Waals pair, or a pair listed twice): the kernels then keep walking the two tables."""
    s5, i5, p5 = vdw
    s6, i6, p6 = ele
    n5, n6 = i5.shape[0], i6.shape[0]
    if n5 == 0 or s5.numel() != s6.numel():
        return None
    dev = i5.device

    def keys(starts, idx):
        counts = (starts[1:] - starts[:-1]).to(torch.int64)
        seg = torch.repeat_interleave(torch.arange(counts.numel(), device=dev, dtype=torch.int64), counts, output_size=idx.shape[0])
        a, b = idx[:, 0].to(torch.int64), idx[:, 1].to(torch.int64)
        return (seg << 40) | (torch.minimum(a, b) << 20) | torch.maximum(a, b)

    k5 = keys(s5, i5)
    order = torch.argsort(k5, stable=True)
    k5s = k5[order]
    if n5 > 1 and bool((k5s[1:] == k5s[:-1]).any()):
        return None
    par = torch.zeros((n5, 5), dtype=torch.float64, device=dev)
    par[:, 0:2] = p5[order]
    if n6:
        k6 = keys(s6, i6)
        pos = torch.searchsorted(k5s, k6)
        if bool((pos >= n5).any()) or bool((k5s[pos.clamp(max=n5 - 1)] != k6).any()) or torch.unique(k6).numel() != n6:
            return None
        par[pos, 2:5] = p6
    return s5, i5[order].contiguous(), par


def _merged_and_ordered(vdw, ele):
    merged = merge_mmff_nonbonded(vdw, ele)
    if merged is not None and pair_order_enabled():
        merged = (merged[0],) + tuple(diagonal_pair_order(*merged))
    return merged


class _GroupList(list):
    """The resident term groups of a MoleculeTermTables, plus what was derived from them once."""

    merged_nonbonded = None


class FlatForcefieldBatch:
    """`n_systems` independent systems with their term tables resident on one GPU.

    Args:
        kind: DG, ETK, MMFF, UFF or QUARTIC.
        atom_starts: (n_systems + 1,) CSR offsets of each system's atoms.
        groups: one ``(starts, idx, par)`` triple per term group of the kind — ``starts`` (n_systems + 1,),
            ``idx`` (n_terms, n_idx) LOCAL atom indices, ``par`` (n_terms, n_par) float64.
        system_mol: optional (n_systems,) int32 (host array or tensor on ``device``).  When given, the term tables are
            stored once per MOLECULE (``starts`` has n_mols + 1 entries) and system s uses row ``system_mol[s]``: the
            conformers of one molecule share their tables (the reference flattens once per unique molecule and
            copies, src/minimizer/bfgs_mmff.cpp:159,195-201).
    """

    def __init__(self, kind: int, atom_starts, groups: Sequence[tuple], device="cuda", system_mol=None):
        if kind not in GROUP_LAYOUT:
            raise ValueError(f"unknown force-field kind {kind}")
        layout = list(GROUP_LAYOUT[kind])
        n_extra = len(groups) - len(layout)
        if n_extra < 0 or (n_extra > 0 and (kind not in (MMFF, UFF) or n_extra > len(CONSTRAINT_LAYOUT))):
            raise ValueError(f"kind {kind} needs {len(layout)} term groups"
                             f"{' (+ up to 4 constraint groups)' if kind in (MMFF, UFF) else ''}, got {len(groups)}")
        layout += CONSTRAINT_LAYOUT[:n_extra]
        self.kind = kind
        self.dim = DIM[kind]
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.atom_starts_host = np.ascontiguousarray(atom_starts, dtype=np.int32)
        self.n_systems = len(self.atom_starts_host) - 1
        if self.n_systems < 0 or np.any(np.diff(self.atom_starts_host) < 0):
            raise ValueError("atom_starts must be a non-decreasing CSR offset array")
        self._keep = [torch.from_numpy(self.atom_starts_host).to(self.device)]
        self._c = _native.FFBatch()
        self._c.kind = kind
        self._c.n_systems = self.n_systems
        self._c.atom_starts = self._keep[0].data_ptr()
        n_rows = self.n_systems
        if system_mol is not None:
            sm = system_mol if isinstance(system_mol, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(system_mol, dtype=np.int32))
            sm = sm.to(device=self.device, dtype=torch.int32).contiguous()
            if sm.numel() != self.n_systems:
                raise ValueError("system_mol must have one entry per system")
            self._keep.append(sm)
            self._c.system_mol = sm.data_ptr()
            n_rows = None  # validated against the tables below
        resident_nonbonded = {}
        for g, ((n_idx, n_par), (starts, idx, par)) in enumerate(zip(layout, groups)):
            if isinstance(starts, torch.Tensor):  # already resident (MoleculeTermTables): validated and ordered there
                t = [starts, idx, par]
                if any(x.device != self.device for x in t):
                    raise ValueError(f"term group {g}: resident tables live on {starts.device}, the batch on {self.device}")
                if n_rows is None:
                    n_rows = starts.numel() - 1
                if starts.numel() != n_rows + 1:
                    raise ValueError(f"term group {g}: inconsistent starts / idx / par sizes")
            else:
                starts = np.ascontiguousarray(starts, dtype=np.int32)
                idx = np.ascontiguousarray(idx, dtype=np.int32).reshape(-1, n_idx)
                par = (np.ascontiguousarray(par, dtype=np.float64).reshape(-1, n_par) if n_par else np.zeros((len(idx), 0)))
                if n_rows is None:
                    n_rows = len(starts) - 1
                if len(starts) != n_rows + 1 or (len(starts) and starts[-1] != len(idx)) or len(par) != len(idx):
                    raise ValueError(f"term group {g}: inconsistent starts / idx / par sizes")
                t = [torch.from_numpy(starts).to(self.device), torch.from_numpy(idx.copy()).to(self.device),
                     torch.from_numpy(np.ascontiguousarray(par)).to(self.device)]
                if g in PAIR_ORDER_GROUPS[kind] and pair_order_enabled():
                    t[1], t[2] = diagonal_pair_order(t[0], t[1], t[2])
            self._keep.extend(t)
            self._c.groups[g].starts = t[0].data_ptr()
            self._c.groups[g].idx = t[1].data_ptr() if t[1].numel() else None
            self._c.groups[g].par = t[2].data_ptr() if t[2].numel() else None
            if kind == MMFF and g in (5, 6):
                resident_nonbonded[g] = tuple(t)
        if kind == MMFF and len(resident_nonbonded) == 2 and mmff_merge_enabled():
            merged = getattr(groups, "merged_nonbonded", False)  # MoleculeTermTables carries it (None = cannot be merged)
            if merged is False:
                merged = _merged_and_ordered(resident_nonbonded[5], resident_nonbonded[6])
            if merged is not None:
                self._keep.extend(merged)
                self._c.groups[11].starts = merged[0].data_ptr()
                self._c.groups[11].idx = merged[1].data_ptr()
                self._c.groups[11].par = merged[2].data_ptr()

    @property
    def n_atoms_total(self) -> int:
        return int(self.atom_starts_host[-1]) if self.n_systems >= 0 and len(self.atom_starts_host) else 0

    def _check_pos(self, pos: torch.Tensor) -> torch.Tensor:
        if not isinstance(pos, torch.Tensor) or not pos.is_cuda or pos.dtype != torch.float64:
            raise ValueError("positions must be a float64 CUDA tensor")
        if pos.device != self.device:
            raise ValueError(f"positions live on {pos.device} but the batch's tables are on {self.device}")
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
        with torch.cuda.device(self.device):  # kernels, scratch and the default stream belong to THIS batch's GPU
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
        with torch.cuda.device(self.device):
            rc = _native.lib().nvmk_ff_gradient(ctypes.byref(self._c), float(w0), float(w1), pos.data_ptr(),
                                                m.data_ptr() if m is not None else None, grad.data_ptr(),
                                                _native.stream_ptr(stream))
        _native.check(rc, "nvmk_ff_gradient")
        return grad

    def minimize(self, pos: torch.Tensor, max_iters: int = 200, grad_tol: float = 1e-4, scale_grads: bool = True,
                 w0: float = 1.0, w1: float = 1.0, active=None, stream=None, restarts: int = 0):
        """BFGS-minimise every (active) system in place.

        Returns ``(energies, statuses, iterations)``; status 0 = converged (reference: BfgsBatchMinimizer::minimize,
        src/minimizer/bfgs_minimize.cu:978-1084; fused kernel bfgs_minimize_permol_kernels.cu:426-745).  ``restarts`` > 0:
        a system that stops at ``max_iters`` is minimised again inside the launch, with a fresh inverse Hessian, that many
        more times (the ETKDG stages' repeatUntilConverged); iterations are the last minimisation's."""
        self._check_pos(pos)
        n = max(self.n_systems, 0)
        energies = torch.zeros(n, dtype=torch.float64, device=self.device)
        statuses = torch.full((n,), -1, dtype=torch.int16, device=self.device)
        iters = torch.zeros(n, dtype=torch.int32, device=self.device)
        m = self._mask(active)
        with torch.cuda.device(self.device):
            rc = _native.lib().nvmk_bfgs_minimize_repeat(ctypes.byref(self._c), self.atom_starts_host.ctypes.data, float(w0),
                                                         float(w1), int(max_iters), int(restarts), float(grad_tol),
                                                         int(bool(scale_grads)), pos.data_ptr(),
                                                         m.data_ptr() if m is not None else None, energies.data_ptr(),
                                                         statuses.data_ptr(), iters.data_ptr(), _native.stream_ptr(stream))
        _native.check(rc, "nvmk_bfgs_minimize")
        return energies, statuses, iters


class MoleculeTermTables:
    """Per-MOLECULE term tables of `kind`, stacked, uploaded and pair-ordered ONCE, for any number of conformer batches.

    ``tables[m][g] = (idx, par)`` as for :func:`stack_molecule_tables`.  The reference flattens once per unique molecule
    and copies into every batch (src/minimizer/bfgs_mmff.cpp:159,195-201); here the tables stay where they are and a
    batch only adds its atom offsets and its system -> molecule map (``FlatForcefieldBatch(..., system_mol=...)``).
    """

    def __init__(self, kind: int, tables: Sequence[Sequence[tuple]], device="cuda"):
        self.kind = kind
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n_mols = len(tables)
        self.groups = _GroupList()
        with torch.cuda.device(self.device):
            for g, (starts, idx, par) in enumerate(stack_molecule_tables(kind, tables)):
                t = [torch.from_numpy(np.ascontiguousarray(starts, dtype=np.int32)).to(self.device),
                     torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int32)).to(self.device),
                     torch.from_numpy(np.ascontiguousarray(par, dtype=np.float64)).to(self.device)]
                if g in PAIR_ORDER_GROUPS[kind] and pair_order_enabled():
                    t[1], t[2] = diagonal_pair_order(t[0], t[1], t[2])
                self.groups.append(tuple(t))
            if kind == MMFF and mmff_merge_enabled():
                self.groups.merged_nonbonded = _merged_and_ordered(self.groups[5], self.groups[6])


def stack_molecule_tables(kind: int, tables: Sequence[Sequence[tuple]]):
    """Per-molecule term tables ``tables[m][g] = (idx, par)`` -> the ``(starts, idx, par)`` groups of a batch whose
    rows are MOLECULES (to be used with ``system_mol``)."""
    layout = GROUP_LAYOUT[kind]
    groups = []
    for g, (n_idx, n_par) in enumerate(layout):
        starts = np.zeros(len(tables) + 1, dtype=np.int32)
        for m, t in enumerate(tables):
            if len(t) != len(layout):
                raise ValueError(f"molecule {m}: kind {kind} needs {len(layout)} term groups, got {len(t)}")
            starts[m + 1] = starts[m] + len(t[g][0])
        idx = (np.concatenate([np.asarray(t[g][0], dtype=np.int32).reshape(-1, n_idx) for t in tables])
               if tables else np.zeros((0, n_idx), dtype=np.int32))
        par = (np.concatenate([np.asarray(t[g][1], dtype=np.float64).reshape(-1, n_par) for t in tables])
               if tables else np.zeros((0, n_par)))
        groups.append((starts, idx, par))
    return groups


def minimize_device_conformers(kind: int, tables, conformers: Device3DResult, max_iters: int, grad_tol: float = 1e-4,
                               stream=None) -> Device3DResult:
    """BFGS-minimise every conformer of a :class:`Device3DResult` without leaving the GPU.

    The counterpart of the reference's ``deviceInput`` path (src/minimizer/bfgs_mmff.cpp:41-328 with
    ``detail::broadcastDeviceInputBatch``; nvmolkit/types.py:197-319): coordinates produced by ETKDG are consumed
    where they are, the term tables are stored once per molecule and shared by its conformers, and the result is a
    new ``Device3DResult`` carrying ``energies`` and ``converged``.  The input result is left untouched."""
    if kind not in (MMFF, UFF):
        raise ValueError("minimize_device_conformers supports the MMFF and UFF kinds")
    n_tables = tables.n_mols if isinstance(tables, MoleculeTermTables) else len(tables)
    if n_tables != conformers.n_mols:
        raise ValueError(f"expected term tables for {conformers.n_mols} molecules, got {n_tables}")
    values = conformers.values.torch()
    device = values.device
    atom_starts = conformers.atom_starts.torch()
    mols = conformers.mol_indices.torch().to(torch.int32)
    if isinstance(tables, MoleculeTermTables):
        if tables.kind != kind:
            raise ValueError("the resident term tables belong to another force field")
        groups = tables.groups
    else:
        groups = stack_molecule_tables(kind, tables)
    batch = FlatForcefieldBatch(kind, atom_starts.cpu().numpy(), groups, device=device, system_mol=mols)
    pos = values.reshape(-1).clone()
    energies, statuses, _ = batch.minimize(pos, max_iters=max_iters, grad_tol=grad_tol, scale_grads=True, stream=stream)
    return Device3DResult(pos.view(-1, 3), atom_starts, conformers.mol_indices, conformers.conf_indices,
                          conformers.gpu_id, conformers.n_mols, energies=energies, converged=(statuses == 0).to(torch.int8))
