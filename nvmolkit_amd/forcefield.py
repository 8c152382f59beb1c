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
import time
import weakref
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
#: pair-term groups whose rows the table builder (csrc/table_build.cpp) re-orders along the diagonals of the pair matrix:
#: RDKit (and every natural builder) emits the O(N^2) pair lists i-major, so 64 consecutive terms share atom i, the 64 lanes
#: of a wavefront add their forces into the SAME three LDS words and the hardware serialises the atomic adds lane by lane;
#: along a diagonal consecutive terms touch distinct atoms on both sides.  Energy and gradient are sums over terms, so only the
#: floating-point summation order changes.  The small 1-2 / 1-3 groups keep the caller's order (ETK groups 2 and 3 are tied to
#: per-system reference distances).  ``NVMK_PAIR_ORDER=input`` / ``NVMK_MMFF_MERGE=0`` switch ordering / merging off (A/B).
PAIR_ORDER_GROUPS = {DG: (0,), ETK: (5,), MMFF: (5, 6), UFF: (4,), QUARTIC: ()}
# MMFF / UFF batches may append up to four constraint groups (distance, position, angle, torsion; include/nvmolkit_amd.h)
CONSTRAINT_LAYOUT = [(2, 3), (1, 5), (3, 3), (4, 3)]


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
        if isinstance(groups, MoleculeTermTables):  # resident: assembled, ordered and (MMFF) merged when they were built
            if system_mol is None:
                raise ValueError("resident per-molecule term tables need system_mol")
            if groups.kind != kind:
                raise ValueError("the resident term tables belong to another force field")
            if groups.device != self.device:
                raise ValueError(f"resident tables live on {groups.device}, the batch on {self.device}")
            tables = groups
        elif kind == QUARTIC:
            if len(groups):
                raise ValueError("the quartic field has no term groups")
            return
        else:
            tables = MoleculeTermTables.from_stacked(kind, groups, self.device, n_rows)
        self._keep.append(tables)
        self._tables = tables
        for g in range(12):
            self._c.groups[g] = tables.view[g]

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

    def _meet_tables(self, stream) -> None:
        """The term tables' uploads precede the kernels of this call, whatever stream it runs on (one hipStreamWaitEvent)."""
        tables = getattr(self, "_tables", None)
        if tables is not None:
            tables.wait(stream)

    @staticmethod
    def _mask(active):
        return None if active is None else active.to(torch.uint8).contiguous()

    def compute_energy(self, pos: torch.Tensor, w0: float = 1.0, w1: float = 1.0, active=None, stream=None) -> torch.Tensor:
        """Per-system energies (reference: BatchedForcefield::computeEnergy)."""
        self._check_pos(pos)
        out = torch.zeros(max(self.n_systems, 0), dtype=torch.float64, device=self.device)
        m = self._mask(active)
        self._meet_tables(stream)
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
        self._meet_tables(stream)
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
        self._meet_tables(stream)
        with torch.cuda.device(self.device):
            rc = _native.lib().nvmk_bfgs_minimize_repeat(ctypes.byref(self._c), self.atom_starts_host.ctypes.data, float(w0),
                                                         float(w1), int(max_iters), int(restarts), float(grad_tol),
                                                         int(bool(scale_grads)), pos.data_ptr(),
                                                         m.data_ptr() if m is not None else None, energies.data_ptr(),
                                                         statuses.data_ptr(), iters.data_ptr(), _native.stream_ptr(stream))
        _native.check(rc, "nvmk_bfgs_minimize")
        return energies, statuses, iters


class MoleculeTermTables:
    """Per-MOLECULE term tables of `kind`, assembled, pair-ordered and uploaded ONCE, for any number of conformer batches.

    ``tables[m][g] = (idx, par)`` as for :func:`stack_molecule_tables`.  The reference flattens once per unique molecule
    and copies into every batch (src/minimizer/bfgs_mmff.cpp:159,195-201); here the tables stay where they are and a
    batch only adds its atom offsets and its system -> molecule map (``FlatForcefieldBatch(..., system_mol=...)``).
    Assembly is the library's (``nvmk_ff_tables_build``: host threads, pinned staging, chunked upload on the current stream
    of ``device``; for MMFF the merged non-bonded table, group 11, is made in the same pass); ``device="cpu"`` builds the
    same tables in host memory for the CPU test-suite.
    """

    def __init__(self, kind: int, tables: Sequence[Sequence[tuple]], device="cuda", preprocessing_threads: int = -1):
        layout = self._start(kind, device, len(GROUP_LAYOUT.get(kind, ())))
        self.n_mols = len(tables)
        n_idx = (ctypes.c_int32 * len(layout))(*[a for a, _ in layout])
        n_par = (ctypes.c_int32 * len(layout))(*[b for _, b in layout])
        terms = (_native.HostTerms * max(self.n_mols * len(layout), 1))()
        keep: list = []
        _native.pyglue().nvmk_py_gather_term_tables(tables, ctypes.addressof(n_idx), ctypes.addressof(n_par), len(layout),
                                                    ctypes.addressof(terms), keep, _native._as_term_array)
        self._build(ctypes.addressof(terms), len(layout), preprocessing_threads)

    @classmethod
    def from_stacked(cls, kind: int, groups: Sequence[tuple], device="cuda", n_rows: int | None = None, preprocessing_threads: int = -1):
        """The same from STACKED groups ``(starts, idx, par)`` (host arrays; ``starts`` has one more entry than there are rows —
        molecules, or systems for a batch without ``system_mol``); MMFF / UFF may append up to four constraint groups."""
        self = cls.__new__(cls)
        layout = self._start(kind, device, len(groups))
        arrays, counts = [], []
        for g, ((n_idx, n_par), (starts, idx, par)) in enumerate(zip(layout, groups)):
            starts = np.ascontiguousarray(starts, dtype=np.int32)
            idx = np.ascontiguousarray(idx, dtype=np.int32).reshape(-1, n_idx)
            par = np.ascontiguousarray(par, dtype=np.float64).reshape(-1, n_par) if n_par else np.zeros((len(idx), 0))
            if n_rows is None:
                n_rows = len(starts) - 1
            if (len(starts) != n_rows + 1 or (len(starts) and (starts[0] != 0 or starts[-1] != len(idx))) or len(par) != len(idx)
                    or np.any(np.diff(starts) < 0)):
                raise ValueError(f"term group {g}: inconsistent starts / idx / par sizes")
            arrays.append((starts, idx, par))
        self.n_mols = int(n_rows or 0)
        terms = np.zeros((self.n_mols, len(layout)), dtype=_HOST_TERMS_DTYPE)
        for g, ((n_idx, n_par), (starts, idx, par)) in enumerate(zip(layout, arrays)):
            if self.n_mols == 0:
                continue
            first = starts[:-1].astype(np.uint64)
            terms["n_terms"][:, g] = np.diff(starts)
            terms["idx_bytes"][:, g] = 4
            terms["idx"][:, g] = np.uint64(idx.ctypes.data) + first * np.uint64(4 * n_idx)
            terms["par"][:, g] = np.uint64(par.ctypes.data) + first * np.uint64(8 * n_par)
        self._build(terms.ctypes.data, len(layout), preprocessing_threads)
        del arrays  # the rows were copied during the build
        return self

    def _start(self, kind: int, device, n_groups: int):
        if kind not in (DG, ETK, MMFF, UFF):
            raise ValueError(f"no term tables for force-field kind {kind}")
        layout = list(GROUP_LAYOUT[kind])
        n_extra = n_groups - len(layout)
        if n_extra < 0 or (n_extra > 0 and (kind not in (MMFF, UFF) or n_extra > len(CONSTRAINT_LAYOUT))):
            raise ValueError(f"kind {kind} needs {len(layout)} term groups"
                             f"{' (+ up to 4 constraint groups)' if kind in (MMFF, UFF) else ''}, got {n_groups}")
        self.kind = kind
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        return layout + CONSTRAINT_LAYOUT[:n_extra]

    def _build(self, terms_address: int, n_groups: int, preprocessing_threads: int) -> None:
        handle = ctypes.c_void_p()
        flags = _native.build_flags()
        if self.device.type == "cuda":
            with torch.cuda.device(self.device):
                rc = _native.lib().nvmk_ff_tables_build(self.kind, terms_address, self.n_mols, n_groups, _native.build_threads(preprocessing_threads), flags,
                                                        _native.stream_ptr(None), ctypes.byref(handle))
        else:
            rc = _native.lib().nvmk_ff_tables_build(self.kind, terms_address, self.n_mols, n_groups, _native.build_threads(preprocessing_threads),
                                                    flags | _native.BUILD_HOST, None, ctypes.byref(handle))
        _native.check(rc, "nvmk_ff_tables_build")
        self._handle = handle
        self._finalizer = weakref.finalize(self, _native.lib().nvmk_ff_tables_free, handle)
        self.view = (_native.FFGroup * 12)()
        _native.check(_native.lib().nvmk_ff_tables_view(handle, ctypes.addressof(self.view), None), "nvmk_ff_tables_view")

    def wait(self, stream=None) -> "MoleculeTermTables":
        """The tables' uploads (on a stream of the build's own) precede what ``stream`` — default: the current stream of the
        tables' device — runs next.  The stream the tables were built under waits already; a consumer on ANOTHER stream calls
        this first (the uploads are asynchronous, so build-on-A / run-on-B would read half-uploaded tables)."""
        if self.device.type == "cuda":
            with torch.cuda.device(self.device):
                _native.check(_native.lib().nvmk_ff_tables_wait(self._handle, _native.stream_ptr(stream)), "nvmk_ff_tables_wait")
        return self


class PendingTermTables:
    """:class:`MoleculeTermTables` under construction on a host thread and a side stream of its own while the caller keeps the
    GPU busy with something else — the MMFF tables of a molecule set while its ETKDG embedding runs, which is how the
    reference hides its per-batch flattening behind the previous batch's kernels (src/minimizer/bfgs_mmff.cpp:139-201).
    ``result()`` joins the thread and makes the caller's current stream wait for the upload."""

    def __init__(self, kind: int, tables, device="cuda", preprocessing_threads: int = -1, after=None):
        """``after``: an object with ``wait(stream)`` — e.g. the :class:`FlatMoleculeSet` of the same molecules, whose own asynchronous
        fill the first ETKDG batch is waiting for: the tables' assembly (seconds of all host threads on a file with peptides)
        starts when that fill is through instead of competing with it."""
        import threading

        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._stream = torch.cuda.Stream(device=self.device)
        self._tables = self._error = None

        def work():
            try:
                # the first step (walking the Python lists into descriptors) holds the GIL for tens of milliseconds: let the
                # caller get into its own blocking library call (which releases the GIL) first, instead of stalling it
                time.sleep(0.003)
                with torch.cuda.device(self.device), torch.cuda.stream(self._stream):
                    if after is not None:
                        after.wait(self._stream)
                    self._tables = MoleculeTermTables(kind, tables, self.device, preprocessing_threads)
            except BaseException as exc:  # noqa: BLE001 - re-raised in result()
                self._error = exc

        self._thread = threading.Thread(target=work, name="nvmk-term-tables")
        self._thread.start()

    def result(self) -> "MoleculeTermTables":
        self._thread.join()
        if self._error is not None:
            raise self._error
        self._tables.wait()  # (the caller's current stream; consumers on other streams wait again for themselves)
        return self._tables


_HOST_TERMS_DTYPE = np.dtype([("n_terms", np.int32), ("idx_bytes", np.int32), ("idx", np.uint64), ("par", np.uint64)])
assert _HOST_TERMS_DTYPE.itemsize == ctypes.sizeof(_native.HostTerms)


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
               if tables and n_par else np.zeros((len(idx), n_par)))
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
    if isinstance(tables, PendingTermTables):
        tables = tables.result()
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
        groups = tables
    else:
        with torch.cuda.device(device):
            groups = MoleculeTermTables(kind, tables, device)
    batch = FlatForcefieldBatch(kind, atom_starts.cpu().numpy(), groups, device=device, system_mol=mols)
    pos = values.reshape(-1).clone()
    energies, statuses, _ = batch.minimize(pos, max_iters=max_iters, grad_tol=grad_tol, scale_grads=True, stream=stream)
    return Device3DResult(pos.view(-1, 3), atom_starts, conformers.mol_indices, conformers.conf_indices,
                          conformers.gpu_id, conformers.n_mols, energies=energies, converged=(statuses == 0).to(torch.int8))
