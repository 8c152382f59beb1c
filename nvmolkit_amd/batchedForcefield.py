"""Object API for batched MMFF / UFF force fields with constraints (reference: nvmolkit/batchedForcefield.py:95-714 over
src/forcefields/forcefield_constraints.cpp:128-232 and the constraint terms of mmff_kernels_device.cuh:663-1036).

``MMFFBatchedForcefield(molecules, ...)`` / ``UFFBatchedForcefield(molecules, ...)`` keep the reference's constructors,
``ff[i].add_*_constraint(...)`` element API, ``compute_energy()`` / ``compute_gradients()`` nested-list results and
``minimize(maxIters, forceTol, output)``.  They need RDKit for typing (the flatteners of mmffOptimization /
uffOptimization).  ``FlatBatchedForcefield(kind, tables, conformers)`` is the same object on flattened term tables and
coordinate arrays — the seam the tests exercise.

Constraints are resolved per conformer when the batch is built, as in the reference: ``relative`` bounds are offsets from
the conformer's current distance / angle / dihedral and position restraints anchor at its current coordinates.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence

import numpy as np
import torch

from nvmolkit_amd.forcefield import CONSTRAINT_LAYOUT, GROUP_LAYOUT, MMFF, UFF, FlatForcefieldBatch
from nvmolkit_amd.types import CoordinateOutput, Device3DResult, HardwareOptions

__all__ = ["FlatBatchedForcefield", "MMFFBatchedForcefield", "UFFBatchedForcefield", "MMFFBatchElement", "UFFBatchElement"]


@dataclass(frozen=True)
class _Restraint:
    """One restraint as the caller stated it: which quantity, on which atoms, and a window [lower, upper] with a force constant.
    ``relative`` windows are offsets from the quantity's value in each conformer; a position restraint has no window, only the
    radius ``upper`` around the atom's position in each conformer."""

    quantity: str          # "distance" | "position" | "angle" | "torsion" — also the order of the four constraint groups
    atoms: tuple
    relative: bool
    lower: float
    upper: float
    force_constant: float


_QUANTITIES = ("distance", "position", "angle", "torsion")


def _normalize_deg(a: float) -> float:
    a = float(np.fmod(a, 360.0))
    if a < -180.0:
        a += 360.0
    elif a > 180.0:
        a -= 360.0
    return a


def _angle_deg(xyz, i, j, k) -> float:
    r1, r2 = xyz[i] - xyz[j], xyz[k] - xyz[j]
    l1, l2 = max(float(r1 @ r1), 1e-5), max(float(r2 @ r2), 1e-5)
    return float(np.degrees(np.arccos(np.clip(float(r1 @ r2) / np.sqrt(l1 * l2), -1.0, 1.0))))


def _dihedral_deg(xyz, i, j, k, l) -> float:  # noqa: E741
    r0, r1, r3 = xyz[i] - xyz[j], xyz[k] - xyz[j], xyz[l] - xyz[k]
    t0, t1 = np.cross(r0, r1), np.cross(-r1, r3)
    t0 = t0 / max(float(np.linalg.norm(t0)), 1e-5)
    t1 = t1 / max(float(np.linalg.norm(t1)), 1e-5)
    cos_phi = float(np.clip(t0 @ t1, -1.0, 1.0))
    m = np.cross(t0, r1)
    return float(np.degrees(-np.arctan2(float(m @ t1) / max(float(np.linalg.norm(m)), 1e-5), cos_phi)))


class _MoleculeRestraints:
    """Everything a caller asked of ONE molecule of the batch.  Adding to it tells the owning batch that its device tables are
    out of date (``notify``); ``rows_for(xyz)`` turns the statements into the four constraint term groups for ONE conformer —
    relative windows are centred on that conformer's current value, position restraints anchor where its atom is now
    (semantics: src/forcefields/forcefield_constraints.cpp:128-232)."""

    def __init__(self, molecule: int, n_atoms: int, notify):
        self.molecule, self.n_atoms, self._notify = molecule, n_atoms, notify
        self.items: list[_Restraint] = []

    def add(self, quantity: str, atoms, relative: bool, lower: float, upper: float, force_constant: float) -> None:
        atoms = tuple(int(a) for a in atoms)
        for a in atoms:
            if not 0 <= a < self.n_atoms:
                raise IndexError(f"molecule {self.molecule} of the batch has {self.n_atoms} atoms: there is no atom {a}")
        self.items.append(_Restraint(quantity, atoms, bool(relative), float(lower), float(upper), float(force_constant)))
        self._notify()

    def __bool__(self) -> bool:
        return bool(self.items)

    def rows_for(self, xyz: np.ndarray):
        rows = {q: [] for q in _QUANTITIES}
        for r in self.items:
            lo, hi = r.lower, r.upper
            if r.quantity == "position":
                rows["position"].append((*r.atoms, *xyz[r.atoms[0]], hi, r.force_constant))
                continue
            if hi < lo:
                raise ValueError({"distance": "Distance constraint maxLen must be >= minLen",
                                  "angle": "Angle constraint maxAngleDeg must be >= minAngleDeg",
                                  "torsion": "Torsion constraint maxDihedralDeg must be >= minDihedralDeg"}[r.quantity])
            if r.quantity == "distance":
                if r.relative:
                    d = float(np.linalg.norm(xyz[r.atoms[0]] - xyz[r.atoms[1]]))
                    lo, hi = max(lo + d, 0.0), max(hi + d, 0.0)
            elif r.quantity == "angle":
                if r.relative:
                    now = _angle_deg(xyz, *r.atoms)
                    lo, hi = lo + now, hi + now
                if not (0.0 <= lo <= 180.0 and 0.0 <= hi <= 180.0):
                    raise ValueError("Angle constraint bounds must be within [0, 180]")
            else:
                if r.relative:
                    now = _dihedral_deg(xyz, *r.atoms)
                    lo, hi = lo + now, hi + now
                lo, hi = _normalize_deg(lo), _normalize_deg(hi)
            rows[r.quantity].append((*r.atoms, lo, hi, r.force_constant))
        groups = []
        for q, (n_idx, n_par) in zip(_QUANTITIES, CONSTRAINT_LAYOUT):
            a = np.array(rows[q], dtype=np.float64).reshape(-1, n_idx + n_par)
            groups.append((a[:, :n_idx].astype(np.int32), a[:, n_idx:]))
        return groups


class _BatchElement:
    """``ff[i]``: one molecule of the batch; constraints added here apply to all of its conformers."""

    def __init__(self, restraints: _MoleculeRestraints):
        self._restraints = restraints

    @property
    def num_atoms(self) -> int:
        return self._restraints.n_atoms

    def add_distance_constraint(self, idx1: int, idx2: int, relative: bool, min_len: float, max_len: float,
                                force_constant: float) -> None:
        self._restraints.add("distance", (idx1, idx2), relative, min_len, max_len, force_constant)

    def add_position_constraint(self, idx: int, max_displ: float, force_constant: float) -> None:
        self._restraints.add("position", (idx,), False, 0.0, max_displ, force_constant)

    def add_angle_constraint(self, idx1: int, idx2: int, idx3: int, relative: bool, min_angle_deg: float, max_angle_deg: float,
                             force_constant: float) -> None:
        self._restraints.add("angle", (idx1, idx2, idx3), relative, min_angle_deg, max_angle_deg, force_constant)

    def add_torsion_constraint(self, idx1: int, idx2: int, idx3: int, idx4: int, relative: bool, min_dihedral_deg: float,
                               max_dihedral_deg: float, force_constant: float) -> None:
        self._restraints.add("torsion", (idx1, idx2, idx3, idx4), relative, min_dihedral_deg, max_dihedral_deg, force_constant)


class MMFFBatchElement(_BatchElement):
    """Per-molecule view of an MMFF batch: ``ff[i]`` (reference: nvmolkit/batchedForcefield.py:291-306)."""


class UFFBatchElement(_BatchElement):
    """Per-molecule view of a UFF batch: ``ff[i]`` (reference: nvmolkit/batchedForcefield.py:309-321)."""


class FlatBatchedForcefield:
    """A batch of molecules, each with its term tables and conformer coordinates, as ONE force-field object.

    Args:
        kind: ``forcefield.MMFF`` or ``forcefield.UFF``.
        tables: ``tables[m]`` = the molecule's base term groups ``[(idx, par), ...]`` (include/nvmolkit_amd.h).
        conformers: ``conformers[m]`` = array (n_confs_m, n_atoms_m, 3); updated in place by :meth:`minimize`.
        gpu_ids: the GPUs :meth:`minimize` deals the conformers over (``HardwareOptions.gpuIds``); energies and gradients
            are evaluated on ``device``, where the object's tables live — the reference's wrapper is built the same way
            (nvmolkit/batchedForcefield.cpp:252,270-272: ``cudaGetDevice`` for the persistent state, ``hwOpts_`` for the
            minimisation).  Empty / one entry: everything on ``device``.
    """

    def __init__(self, kind: int, tables: Sequence, conformers: Sequence[np.ndarray], device="cuda", gpu_ids: Sequence[int] | None = None):
        if kind not in (MMFF, UFF):
            raise ValueError("kind must be forcefield.MMFF or forcefield.UFF")
        if len(tables) != len(conformers):
            raise ValueError("one term table and one conformer array per molecule")
        self.kind = kind
        self._element_type = MMFFBatchElement if kind == MMFF else UFFBatchElement
        self.device = torch.device(device)
        self._tables = list(tables)
        self._conformers = [np.ascontiguousarray(c, dtype=np.float64).reshape(len(c), -1, 3) for c in conformers]
        n = len(self._tables)
        # a counter that every added restraint advances; the device tables remember the count they were made at
        self._edits = 0
        self._restraints = [_MoleculeRestraints(m, int(c.shape[1]), self._edited) for m, c in enumerate(self._conformers)]
        self._device_tables = None   # (edit count, FlatForcefieldBatch)
        self._gpu_ids = [int(g) for g in gpu_ids] if gpu_ids else []
        self._shards = None          # (edit count, [(gpu, systems of the shard, FlatForcefieldBatch), ...]) of a minimisation over several GPUs
        self.num_molecules = n
        self.data_dim = 3

    def _edited(self) -> None:
        self._edits += 1

    # ---- container protocol ----
    def __len__(self) -> int:
        return self.num_molecules

    def __getitem__(self, idx: int) -> _BatchElement:
        if not 0 <= idx < self.num_molecules:
            raise IndexError(f"the batch holds {self.num_molecules} molecules: there is no molecule {idx}")
        return self._element_type(self._restraints[idx])

    # ---- device tables: made on first use, made again after a restraint was added ----
    def _stack(self, systems, device) -> FlatForcefieldBatch:
        """The device tables of ``systems`` (a list of (molecule, conformer)) made from the CURRENT coordinates."""
        sizes = np.array([self._restraints[m].n_atoms for m, _ in systems], dtype=np.int64)
        atom_starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        restrained = any(self._restraints)
        layout = list(GROUP_LAYOUT[self.kind]) + (list(CONSTRAINT_LAYOUT) if restrained else [])
        per_system = [list(self._tables[m]) + (self._restraints[m].rows_for(self._conformers[m][k]) if restrained else [])
                      for m, k in systems]
        stacked = []
        for g, (n_idx, n_par) in enumerate(layout):
            counts = [len(groups[g][0]) for groups in per_system]
            idx = (np.concatenate([np.asarray(groups[g][0], dtype=np.int32).reshape(-1, n_idx) for groups in per_system])
                   if per_system else np.zeros((0, n_idx), dtype=np.int32))
            par = (np.concatenate([np.asarray(groups[g][1], dtype=np.float64).reshape(-1, n_par) for groups in per_system])
                   if per_system else np.zeros((0, n_par)))
            stacked.append((np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), idx, par))
        return FlatForcefieldBatch(self.kind, atom_starts, stacked, device=device)

    def rebuild(self) -> None:
        """Make the device tables again from the CURRENT coordinates (relative windows and position anchors move with them)."""
        self._systems = [(m, k) for m, c in enumerate(self._conformers) for k in range(len(c))]
        sizes = np.array([self._restraints[m].n_atoms for m, _ in self._systems], dtype=np.int64)
        self._atom_starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        self._device_tables = (self._edits, self._stack(self._systems, self.device))
        self._shards = None

    @property
    def _batch(self) -> FlatForcefieldBatch:
        if self._device_tables is None or self._device_tables[0] != self._edits:
            self.rebuild()
        return self._device_tables[1]

    def _positions(self) -> torch.Tensor:
        flat = (np.concatenate([self._conformers[m][k].reshape(-1) for m, k in self._systems]) if self._systems
                else np.zeros(0))
        return torch.from_numpy(flat).to(self.device)

    def _nest(self, values):
        out = [[] for _ in range(self.num_molecules)]
        for (m, _), v in zip(self._systems, values):
            out[m].append(v)
        return out

    # ---- evaluation ----
    def compute_energy(self) -> list[list[float]]:
        """``result[mol][conf]``: one energy per conformer."""
        if self.num_molecules == 0:
            return []
        batch = self._batch
        return self._nest([float(e) for e in batch.compute_energy(self._positions()).cpu().numpy()])

    def compute_gradients(self) -> list[list[list[float]]]:
        """``result[mol][conf]``: the flattened ``[x0, y0, z0, ...]`` gradient of every conformer."""
        if self.num_molecules == 0:
            return []
        batch = self._batch
        g = batch.compute_gradient(self._positions()).cpu().numpy()
        return self._nest([g[3 * self._atom_starts[s]:3 * self._atom_starts[s + 1]].tolist() for s in range(len(self._systems))])

    def minimize(self, maxIters: int | None = None, forceTol: float = 1e-4,
                 output: CoordinateOutput = CoordinateOutput.RDKIT_CONFORMERS, target_gpu: int | None = None,
                 targetGpu: int | None = None):
        """BFGS-minimise every conformer.  ``RDKIT_CONFORMERS``: coordinates are written back (into the stored arrays, and
        into RDKit conformers when the batch was built from molecules) and ``(energies, converged)`` nested lists are
        returned; ``DEVICE``: a :class:`Device3DResult` and nothing is written back.  ``maxIters`` defaults to the
        reference's 200 (MMFF) / 1000 (UFF) (nvmolkit/batchedForcefield.py:565-571,681-687); ``target_gpu`` is the reference's
        keyword, ``targetGpu`` (the spelling of the optimise drivers) is accepted as well."""
        if maxIters is None:
            maxIters = 200 if self.kind == MMFF else 1000
        if target_gpu is not None and targetGpu is not None and int(target_gpu) != int(targetGpu):
            raise ValueError("target_gpu and targetGpu disagree")
        targetGpu = target_gpu if target_gpu is not None else targetGpu
        if self.num_molecules == 0:
            if output == CoordinateOutput.DEVICE:
                raise ValueError("minimize(output=DEVICE) requires at least one molecule")
            return [], []
        batch = self._batch
        pos = self._positions()
        if len(self._gpu_ids) > 1:
            energies, statuses = self._minimize_over_gpus(pos, int(maxIters), float(forceTol))
        else:
            energies, statuses, _ = batch.minimize(pos, max_iters=int(maxIters), grad_tol=float(forceTol), scale_grads=True)
        if output == CoordinateOutput.DEVICE:
            gpu = self.device.index if self.device.index is not None else torch.cuda.current_device()
            if targetGpu is not None and int(targetGpu) >= 0 and int(targetGpu) != gpu:
                name = "MMFFBatchedForcefield" if self.kind == MMFF else "UFFBatchedForcefield"
                raise ValueError(f"{name}.minimize(output=DEVICE) does not support target_gpu != wrapper GPU "
                                 f"(target_gpu {int(targetGpu)}, wrapper GPU {gpu}); use the optimise drivers' targetGpu "
                                 "for cross-GPU consolidation")  # wording: nvmolkit/batchedForcefield.cpp:283,414
            i32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.int32)).to(self.device)  # noqa: E731
            return Device3DResult(pos.view(-1, 3), i32(self._atom_starts), i32([m for m, _ in self._systems]),
                                  i32([k for _, k in self._systems]), gpu, self.num_molecules, energies=energies,
                                  converged=(statuses == 0).to(torch.int8))
        out = pos.cpu().numpy()
        for s, (m, k) in enumerate(self._systems):
            self._conformers[m][k] = out[3 * self._atom_starts[s]:3 * self._atom_starts[s + 1]].reshape(-1, 3)
        self._write_back()
        return (self._nest([float(e) for e in energies.cpu().numpy()]),
                self._nest([bool(c) for c in (statuses == 0).cpu().numpy()]))

    def _minimize_over_gpus(self, pos: torch.Tensor, max_iters: int, grad_tol: float):
        """The minimisation of :meth:`minimize` with the conformers dealt over ``gpu_ids`` by modelled cost (largest first, as the
        optimise drivers deal molecules: distributed.molecule_owners_by_cost), one host thread per entry, no collective; ``pos``
        (on ``self.device``) receives the minimised coordinates.  A system's result does not depend on which GPU or beside which
        other systems it ran (its size class is a function of its size), so the results equal the one-GPU call's bit for bit.
        Reference: nvmolkit/batchedForcefield.cpp:270-272 hands ``hwOpts_`` to MMFFMinimizeMoleculesConfs, whose threads take
        batches per GPU (src/minimizer/bfgs_mmff.cpp:139-201)."""
        from concurrent.futures import ThreadPoolExecutor

        from nvmolkit_amd.distributed import molecule_owners_by_cost

        if self._shards is None or self._shards[0] != self._edits:
            sizes = np.diff(self._atom_starts)
            owner, _ = molecule_owners_by_cost(sizes, len(self._gpu_ids))
            shards = []
            for r, gpu in enumerate(self._gpu_ids):
                members = np.flatnonzero(owner == r)
                if len(members):
                    shards.append((gpu, members, self._stack([self._systems[s] for s in members], torch.device("cuda", gpu))))
            self._shards = (self._edits, shards)
        starts = self._atom_starts.astype(np.int64)
        pos3 = pos.view(-1, 3)

        # every shard's coordinates are gathered on the object's GPU first; a thread then copies its own to its GPU, minimises
        # there on that device's current stream and brings coordinates, energies and statuses back
        gathered = []
        for _, members, _ in self._shards[1]:
            rows = torch.from_numpy(np.concatenate([np.arange(starts[s], starts[s + 1]) for s in members])).to(self.device)
            gathered.append((rows, pos3[rows]))
        torch.cuda.synchronize(self.device)

        def run(k):
            gpu, _, batch = self._shards[1][k]
            with torch.cuda.device(gpu):
                local = gathered[k][1].to(batch.device).reshape(-1)
                e, st, _ = batch.minimize(local, max_iters=max_iters, grad_tol=grad_tol, scale_grads=True)
                out = (local.view(-1, 3).to(self.device), e.to(self.device), st.to(self.device))
                torch.cuda.synchronize(gpu)
            torch.cuda.synchronize(self.device)
            return out

        with ThreadPoolExecutor(max_workers=len(self._shards[1])) as pool:
            done = list(pool.map(run, range(len(self._shards[1]))))
        energies = torch.empty(len(self._systems), dtype=torch.float64, device=self.device)
        statuses = None
        for (_, members, _), (rows, _), (xyz, e, st) in zip(self._shards[1], gathered, done):
            pos3[rows] = xyz
            where = torch.from_numpy(members).to(self.device)
            energies[where] = e
            if statuses is None:
                statuses = torch.empty(len(self._systems), dtype=st.dtype, device=self.device)
            statuses[where] = st
        return energies, statuses

    def _write_back(self) -> None:  # overridden by the RDKit-backed classes
        pass


class _RdkitBacked(FlatBatchedForcefield):
    def _init_from_molecules(self, kind, molecules, flatten, hardwareOptions):
        none_idx = [i for i, m in enumerate(molecules) if m is None]
        if none_idx:
            raise ValueError(f"None at indices {none_idx}", {"none": none_idx, "no_params": []})
        self._molecules = list(molecules)
        self._hardware_options = hardwareOptions if hardwareOptions is not None else HardwareOptions()
        self._conf_ids = [[c.GetId() for c in m.GetConformers()] for m in molecules]
        conformers = [np.stack([np.asarray(m.GetConformer(cid).GetPositions(), dtype=np.float64) for cid in ids])
                      if ids else np.zeros((0, m.GetNumAtoms(), 3)) for m, ids in zip(molecules, self._conf_ids)]
        tables = [flatten(i, ids[0] if ids else -1) for i, ids in enumerate(self._conf_ids)]
        gpu_ids = self._hardware_options.gpuIds
        super().__init__(kind, tables, conformers, device=torch.device("cuda", gpu_ids[0] if gpu_ids else torch.cuda.current_device()),
                         gpu_ids=gpu_ids)

    def _write_back(self) -> None:
        for m, ids, confs in zip(self._molecules, self._conf_ids, self._conformers):
            for cid, xyz in zip(ids, confs):
                conf = m.GetConformer(cid)
                if hasattr(conf, "SetPositions"):
                    conf.SetPositions(np.ascontiguousarray(xyz))
                else:
                    from rdkit.Geometry import Point3D

                    for a, (x, y, z) in enumerate(xyz):
                        conf.SetAtomPosition(a, Point3D(float(x), float(y), float(z)))


def _per_mol(value, n: int, name: str):
    if isinstance(value, Sequence) and not isinstance(value, (str, bytes)):
        if len(value) != n:
            raise ValueError(f"Expected {n} values for {name}, got {len(value)}")
        return list(value)
    return [value] * n


class MMFFBatchedForcefield(_RdkitBacked):
    """MMFF94 batch over RDKit molecules (reference: nvmolkit/batchedForcefield.py:443-598).  Needs RDKit."""

    def __init__(self, molecules, properties=None, nonBondedThreshold=100.0, ignoreInterfragInteractions=True,
                 hardwareOptions: HardwareOptions | None = None):
        from nvmolkit_amd.mmffOptimization import flatten_mmff_from_rdkit

        n = len(molecules)
        thr, frag = _per_mol(nonBondedThreshold, n, "nonBondedThreshold"), _per_mol(ignoreInterfragInteractions, n, "ignoreInterfragInteractions")
        props = _per_mol(properties, n, "properties") if isinstance(properties, (list, tuple)) else [properties] * n

        def flatten(i, cid):
            from rdkit.Chem import rdForceFieldHelpers as ffh

            p = props[i] if props[i] is not None else ffh.MMFFGetMoleculeProperties(molecules[i])
            if p is None:
                raise ValueError(f"lacking MMFF atom types at indices [{i}]", {"none": [], "no_params": [i]})
            return flatten_mmff_from_rdkit(molecules[i], p, cid, float(thr[i]), bool(frag[i]))

        self._init_from_molecules(MMFF, molecules, flatten, hardwareOptions)


class UFFBatchedForcefield(_RdkitBacked):
    """UFF batch over RDKit molecules (reference: nvmolkit/batchedForcefield.py:601-714).  Needs RDKit."""

    def __init__(self, molecules, vdwThreshold=10.0, ignoreInterfragInteractions=True,
                 hardwareOptions: HardwareOptions | None = None):
        from nvmolkit_amd.uffOptimization import flatten_uff_from_rdkit

        n = len(molecules)
        thr, frag = _per_mol(vdwThreshold, n, "vdwThreshold"), _per_mol(ignoreInterfragInteractions, n, "ignoreInterfragInteractions")
        self._init_from_molecules(UFF, molecules,
                                  lambda i, cid: flatten_uff_from_rdkit(molecules[i], cid, float(thr[i]), bool(frag[i])),
                                  hardwareOptions)
