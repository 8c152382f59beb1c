"""Batched ETKDG conformer embedding on the GPU (reference API: nvmolkit/embedMolecules.py:55-158).

Two entry points:
  * :func:`EmbedMolecules` — the reference's signature, taking RDKit molecules.  All chemistry perception
    (bounds matrix, chiral sets, experimental torsions) is RDKit's, so this adapter needs RDKit.
  * :func:`embed_flat` — the flattened-term seam the kernels actually consume; what the adapter feeds, and what
    the test-suite and benchmarks drive directly.
"""

from __future__ import annotations

import ctypes
import time
import weakref
from dataclasses import dataclass, field
from typing import Sequence

import numpy as np
import torch

from nvmolkit_amd import _native
from nvmolkit_amd.types import CoordinateOutput, Device3DResult, HardwareOptions

N_STAGES = _native.ETKDG_N_STAGES
STAGE_NAMES = ["coordgen", "first minimization", "tetrahedral check", "first chiral check", "fourth dimension minimization",
               "ETK minimization", "double bond geometry", "final chiral check", "chiral distance matrix",
               "chiral centre volume", "double bond stereo"]


def stage_timings() -> list[dict]:
    """Per-stage wall clock of the last ETKDG call that ran with ``NVMK_ETKDG_TIMING=1`` (``_native.options(NVMK_ETKDG_TIMING="1")``):
    one dict per row — the pipeline's stages under the reference's names, then the host work between the stages and the
    whole call — with total / min / max milliseconds and the number of batches (the reference's debug-mode table,
    src/etkdg_impl.cpp:161-200)."""
    import ctypes

    rows = len(STAGE_NAMES) + 2
    total, lo, hi = (np.zeros(rows) for _ in range(3))
    calls = np.zeros(rows, dtype=np.int32)
    names = (ctypes.c_char_p * rows)()
    _native.check(_native.lib().nvmk_etkdg_stage_timings(total.ctypes.data, lo.ctypes.data, hi.ctypes.data, calls.ctypes.data, rows,
                                                         ctypes.cast(names, ctypes.c_void_p)), "nvmk_etkdg_stage_timings")
    return [{"stage": names[r].decode(), "total_ms": float(total[r]), "min_ms": float(lo[r]), "max_ms": float(hi[r]), "calls": int(calls[r])}
            for r in range(rows)]


def format_stage_timings(rows: list[dict] | None = None) -> str:
    """The table as the reference's driver prints it."""
    rows = stage_timings() if rows is None else rows
    out = [f"{'Stage Name':<66}{'Total (ms)':>12}{'Avg (ms)':>12}{'Min (ms)':>12}{'Max (ms)':>12}{'Calls':>8}", "-" * 122]
    for r in rows:
        avg = r["total_ms"] / max(r["calls"], 1)
        out.append(f"{r['stage']:<66}{r['total_ms']:>12.3f}{avg:>12.3f}{r['min_ms']:>12.3f}{r['max_ms']:>12.3f}{r['calls']:>8d}")
    return "\n".join(out)


class StereoChecks:
    """A molecule's stereo checks as three arrays — ``kind`` (n,) int32, ``idx`` (n, 5) int32 (unused slots 0), ``par`` (n, 2)
    float64 — that still reads like the list of ``(kind, idx, par)`` tuples ``FlatMolecule.checks`` is documented as (len,
    iteration, indexing).  The table builder's glue copies the arrays with three memcpys instead of walking ~30 tuples per
    molecule under the GIL (half of the descriptor walk's 67 ms per 10 000 molecules)."""

    __slots__ = ("kind", "idx", "par", "n_idx", "n_par")

    def __init__(self, checks=()):
        checks = list(checks)
        n = len(checks)
        self.kind = np.fromiter((c[0] for c in checks), dtype=np.int32, count=n)
        self.idx = np.zeros((n, 5), dtype=np.int32)
        self.par = np.zeros((n, 2), dtype=np.float64)
        self.n_idx = np.fromiter((len(c[1]) for c in checks), dtype=np.int8, count=n)  # (what a tuple held: iteration gives it back)
        self.n_par = np.fromiter((len(c[2]) for c in checks), dtype=np.int8, count=n)
        for k, (_, idx, par) in enumerate(checks):
            if len(idx) > 5 or len(par) > 2:
                raise ValueError("a stereo check has at most 5 indices and 2 parameters")
            self.idx[k, :len(idx)] = idx
            self.par[k, :len(par)] = par

    def __len__(self) -> int:
        return len(self.kind)

    def __getitem__(self, k):
        return (int(self.kind[k]), tuple(int(x) for x in self.idx[k, :self.n_idx[k]]), tuple(float(x) for x in self.par[k, :self.n_par[k]]))

    def __iter__(self):
        return (self[k] for k in range(len(self)))

    def __getstate__(self):
        return {name: getattr(self, name) for name in self.__slots__}

    def __setstate__(self, state):
        for name, value in state.items():
            setattr(self, name, value)


@dataclass
class FlatMolecule:
    """One molecule in flattened form: term groups with LOCAL atom indices (layouts: include/nvmolkit_amd.h)."""

    n_atoms: int
    dg: Sequence[tuple]                       # 3 x (idx (n, n_idx), par (n, n_par))
    etk: Sequence[tuple] | None = None        # 6 x (idx, par) or None when the ETK stage is off
    checks: Sequence[tuple] = field(default_factory=list)  # (kind, idx[<= 5], par[<= 2]) tuples, or a StereoChecks (arrays: faster to hand over)
    num_impropers: int = 0


class FlatMoleculeSet:
    """Unique molecules resident on the device, ready for :func:`embed_flat`.

    The tables are assembled by the library (``nvmk_etkdg_molset_build``: host threads concatenate the molecules' term groups,
    bring the pair tables into the kernels' order and upload chunk by chunk through pinned staging, on the current stream of
    ``device``) — the counterpart of the reference's host flattening on ``preprocessingThreads`` threads
    (src/etkdg.cpp:175-191), and like it part of what an ``EmbedMolecules`` call costs.  The Python side only hands over
    pointers into the molecules' own arrays (``_native.pyglue``).  ``device="cpu"`` assembles the same tables in host memory
    (no GPU involved; the CPU test-suite reads them back through ``self.c``).

    ``asynchronous`` (default on a GPU): the constructor returns once the tables are PLANNED — sizes known, device block
    allocated — and the library's threads fill and upload the rows in molecule order while the caller goes on;
    ``nvmk_etkdg_embed`` waits before every batch for the rows of that batch's molecules, so the tables of batch k + 1 are
    assembled while batch k is on the GPU (the reference's structure: src/etkdg.cpp:175-191,211-240).  :meth:`wait` blocks until
    every row is uploaded (and raises what the fill may have failed with); anything that reads the tables directly calls it first."""

    def __init__(self, mols: Sequence[FlatMolecule], device="cuda", preprocessing_threads: int = -1, asynchronous: bool = True):
        t0 = time.perf_counter()
        self.device = torch.device(device)
        self.mols = list(mols)
        n = len(self.mols)
        self.n_atoms = np.fromiter((m.n_atoms for m in self.mols), dtype=np.int32, count=n)
        # (capacity of the arrays the glue copies tuple-style checks into; a StereoChecks is referenced where it lies)
        n_checks = sum(len(m.checks) for m in self.mols if not isinstance(m.checks, StereoChecks))
        descs = (_native.FlatMoleculeDesc * max(n, 1))()
        kinds, idx, par = np.empty(n_checks, np.int32), np.empty((n_checks, 5), np.int32), np.empty((n_checks, 2), np.float64)
        keep: list = []
        _native.pyglue().nvmk_py_gather_flat_molecules(self.mols, ctypes.addressof(descs), kinds.ctypes.data, idx.ctypes.data,
                                                       par.ctypes.data, n_checks, keep, _native._as_term_array)
        handle = ctypes.c_void_p()
        flags = _native.build_flags()
        t1 = time.perf_counter()
        if self.device.type == "cuda":
            if asynchronous:
                flags |= _native.BUILD_ASYNC
            # (the library's threads read the molecules' arrays and these descriptors until the fill is through)
            self._alive = (descs, kinds, idx, par, keep)
            with torch.cuda.device(self.device):
                rc = _native.lib().nvmk_etkdg_molset_build(ctypes.addressof(descs), n, _native.build_threads(preprocessing_threads), flags,
                                                           _native.stream_ptr(None), ctypes.byref(handle))
        else:
            rc = _native.lib().nvmk_etkdg_molset_build(ctypes.addressof(descs), n, _native.build_threads(preprocessing_threads),
                                                       flags | _native.BUILD_HOST, None, ctypes.byref(handle))
        _native.check(rc, "nvmk_etkdg_molset_build")
        self._handle = handle
        self._finalizer = weakref.finalize(self, _native.lib().nvmk_etkdg_molset_free, handle)
        self.c = _native.EtkdgMolset()
        _native.check(_native.lib().nvmk_etkdg_molset_view(handle, ctypes.byref(self.c)), "nvmk_etkdg_molset_view")
        self.has_etk = bool(self.c.h_etk_d12_counts)
        #: host seconds of the two steps: descriptors from the Python objects (GIL held), the library's assembly + upload calls
        self.timings = {"gather_seconds": t1 - t0, "build_seconds": time.perf_counter() - t1}

    def wait(self, stream=None) -> "FlatMoleculeSet":
        """Every row uploaded before what ``stream`` (default: the current one) runs next; raises if the fill failed."""
        if self.device.type == "cuda":
            with torch.cuda.device(self.device):
                _native.check(_native.lib().nvmk_etkdg_molset_wait(self._handle, -1, _native.stream_ptr(stream)), "nvmk_etkdg_molset_wait")
        return self


# Conformer attempts per launch when the caller does not choose (-1, as HardwareOptions.batchSize).  One wave (small systems) or
# one four-wave workgroup (larger ones) per attempt, 2048 / 512 of them resident: a batch must be several times that, or the
# tail of its launches — a few long minimisations on an almost idle chip — dominates.  The reference's default is 500.
# Concurrent batches (HardwareOptions.batchesPerGpu, -1 = this default) overlap one batch's tails with the other's bulk, but a
# single batch at a time keeps a seeded run reproducible bit for bit (with several, the scheduler hands out attempts in the
# order the batches finish), and at 16384 attempts it is as fast.  Round 3 sweep, 10 000 molecules x 10 conformers, ETKDG
# conformers/s (profiles/r03_conformers/batch_sweep_wave176.jsonl): 8192 x 1 37.5k, 16384 x 1 40.5k / 40.2k, 24576 x 1 38.4k,
# 32768 x 1 36.8k, 4096 x 2 41.0k, 8192 x 2 38.5k, 16384 x 2 37.1k, 4096 x 3 40.6k, 8192 x 3 41.7k.
# Round 6, after the persistent per-XCD queues, the team classes and the per-batch table pipeline, the same job
# (profiles/r06_conformers/batch_size_sweep.txt, candidates alternating on one box): 6 x 16 667 ETKDG 1.875 / 1.879 s, 3 x 33 334
# 1.835 / 1.831 s, 2 x 50 000 1.821 / 1.815 s, 1 x 100 000 2.28 s; end to end 3417 / 3519, 3562 / 3552, 3595 / 3554 mol/s; the whole
# ChEMBL file 16.7 / 16.4, 15.8 / 16.0, 16.0 / 16.1 s (end to end within the boxes' noise).  Three batches: most of the gain, and
# the first batch still waits for a third of the molecule set's rows only.
AUTO_BATCH_SIZE = 32768
AUTO_BATCHES_PER_GPU = 1


def auto_batch_size(n_attempts: int) -> int:
    """The automatic batch size for a job of ``n_attempts`` first-round attempts: about AUTO_BATCH_SIZE, but EQUAL batches — the
    first round of 10 000 molecules x 10 conformers was 6 x 16 384 + 1696 when batches were 16 384 (3 x 32 768 + 1696 now), and
    that last batch is a launch-latency-bound runt that costs as much wall as a quarter of a full one.  Six batches of 16 667 instead: ETKDG 1.99 -> 1.89 s, 3440 -> 3565 mol/s
    (seven of 14 286: 3557; profiles/r05_conformers/ab_equal_batches.txt).  The scheduler's hand-out order does not depend on where
    the batches are cut (reference sequences: tests/test_scheduler.py)."""
    if n_attempts <= AUTO_BATCH_SIZE:
        return AUTO_BATCH_SIZE
    n_batches = max(1, round(n_attempts / AUTO_BATCH_SIZE))
    return -(-n_attempts // n_batches)


@dataclass
class FlatEmbedResult:
    coords: torch.Tensor            # flat float64, conformer c of molecule m at slot_starts[m] + 3 * c * n_atoms[m]
    conf_counts: np.ndarray         # conformers produced per molecule
    slot_starts: np.ndarray
    stage_failures: np.ndarray      # total failures per stage (STAGE_NAMES)
    n_atoms: np.ndarray

    def conformers(self, m: int) -> torch.Tensor:
        """(n_confs, n_atoms, 3) coordinates of molecule m."""
        n = int(self.n_atoms[m])
        k = int(self.conf_counts[m])
        s = int(self.slot_starts[m])
        return self.coords[s:s + 3 * n * k].reshape(k, n, 3)

    def to_device_result(self) -> Device3DResult:
        """Compact the per-molecule slots into the flat CSR form of :class:`Device3DResult` on the same GPU
        (reference: DeviceCoordCollector / finalizeOnTarget, src/conformer/device_coord_collector.cpp:30-150)."""
        dev = self.coords.device
        counts = self.conf_counts.astype(np.int64)
        n_atoms = self.n_atoms.astype(np.int64)
        mol_of_conf = np.repeat(np.arange(len(counts)), counts)
        conf_of_conf = np.arange(len(mol_of_conf)) - np.repeat(np.cumsum(counts) - counts, counts)
        sizes = n_atoms[mol_of_conf]
        atom_starts = np.zeros(len(sizes) + 1, dtype=np.int64)
        atom_starts[1:] = np.cumsum(sizes)
        src_row0 = self.slot_starts[mol_of_conf] // 3 + conf_of_conf * sizes  # first source row of every conformer
        sizes_t = torch.from_numpy(sizes).to(dev)
        rows = (torch.repeat_interleave(torch.from_numpy(src_row0 - atom_starts[:-1]).to(dev), sizes_t) +
                torch.arange(int(atom_starts[-1]), device=dev))
        values = self.coords.view(-1, 3)[rows]
        i32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.int32)).to(dev)  # noqa: E731
        return Device3DResult(values, i32(atom_starts), i32(mol_of_conf), i32(conf_of_conf),
                              dev.index if dev.index is not None else torch.cuda.current_device(), len(counts))


def embed_flat(molset: FlatMoleculeSet, confs_per_molecule: int = 1, max_iterations: int = -1, batch_size: int = -1,
               use_exp_torsions: bool = True, use_basic_knowledge: bool = True, enforce_chirality: bool = True,
               box_size_mult: float = 2.0, force_tol: float = 1e-3, seed: int = 42, stream=None,
               output: CoordinateOutput = CoordinateOutput.RDKIT_CONFORMERS, batches_per_gpu: int = -1,
               prune_rms_thresh: float = -1.0, prune_atom_subsets=None, prune_self_matches=None):
    """ETKDG on flattened molecules (reference pipeline: src/etkdg.cpp:90-484 downstream of RDKit).

    Returns a :class:`FlatEmbedResult`, or with ``output=CoordinateOutput.DEVICE`` a :class:`Device3DResult` that the
    MMFF / UFF ``optimize_device`` entry points consume without a host round trip.

    ``prune_rms_thresh`` > 0 (EmbedParameters.pruneRmsThresh) prunes the conformers of every molecule on the GPU after the
    embedding (``conformerRmsd.prune_conformers``; ``prune_atom_subsets[m]`` = atom indices for onlyHeavyAtomsForRMS, or
    ``prune_self_matches[m]`` = the molecule's (K, L) self matches for useSymmetryForPruning, e.g. ``SmilesSet.self_matches``) and
    needs ``output=DEVICE`` here — the reference prunes on the CPU and therefore only with RDKit conformer output.

    ``batches_per_gpu`` > 1 runs that many batches concurrently on their own streams (HardwareOptions.batchesPerGpu); -1
    (default) = ``AUTO_BATCHES_PER_GPU`` when the work spans more than two batches, else one batch at a time.  With one
    batch at a time a seed reproduces the same conformers bit for bit; with several, which random start a molecule's n-th
    attempt gets depends on the order the batches finish (as in the reference)."""
    sptr = _native.stream_ptr(stream)
    if confs_per_molecule <= 0:
        raise ValueError("confsPerMolecule must be greater than 0")
    n_atoms = molset.n_atoms
    if max_iterations == -1:  # src/etkdg.cpp:71-85,195-197: 10 x the largest molecule
        max_iterations = 10 * int(n_atoms.max()) if len(n_atoms) else 1
    if max_iterations <= 0:
        raise ValueError("maxIterations must be greater than 0 (or -1 for automatic)")
    if prune_rms_thresh > 0.0 and output != CoordinateOutput.DEVICE:
        raise ValueError("prune_rms_thresh needs output=CoordinateOutput.DEVICE in embed_flat")
    etk_on = (use_exp_torsions or use_basic_knowledge)
    if etk_on and not molset.has_etk:
        raise ValueError("the ETK stage needs ETK term groups on every molecule")
    prm = _native.EtkdgParams()
    prm.confs_per_mol = int(confs_per_molecule)
    prm.max_iterations = int(max_iterations)
    prm.batch_size = int(batch_size) if batch_size > 0 else auto_batch_size(int(len(n_atoms)) * int(confs_per_molecule))
    prm.use_exp_torsions = int(bool(use_exp_torsions))
    prm.use_basic_knowledge = int(bool(use_basic_knowledge))
    prm.enforce_chirality = int(bool(enforce_chirality))
    prm.box_size = 5.0 * box_size_mult if box_size_mult > 0 else -box_size_mult
    prm.force_tol = float(force_tol)
    prm.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    n_attempts = int(len(n_atoms)) * int(confs_per_molecule)
    prm.batches_per_gpu = (int(batches_per_gpu) if batches_per_gpu > 0
                           else (AUTO_BATCHES_PER_GPU if n_attempts > 2 * prm.batch_size else 1))
    slot_starts = np.zeros(len(n_atoms) + 1, dtype=np.int64)
    slot_starts[1:] = np.cumsum(n_atoms.astype(np.int64) * confs_per_molecule * 3)
    coords = torch.zeros(int(slot_starts[-1]), dtype=torch.float64, device=molset.device)
    counts = np.zeros(len(n_atoms), dtype=np.int32)
    fails = np.zeros(N_STAGES, dtype=np.int32)
    with torch.cuda.device(molset.device):
        rc = _native.lib().nvmk_etkdg_embed(ctypes.byref(molset.c), ctypes.byref(prm), coords.data_ptr(), counts.ctypes.data,
                                            fails.ctypes.data, sptr)
    _native.check(rc, "nvmk_etkdg_embed")
    res = FlatEmbedResult(coords, counts, slot_starts[:-1], fails, n_atoms)
    if output != CoordinateOutput.DEVICE:
        return res
    dev = res.to_device_result()
    if prune_rms_thresh > 0.0:
        from nvmolkit_amd.conformerRmsd import prune_conformers

        dev = prune_conformers(dev, float(prune_rms_thresh), prune_atom_subsets, prune_self_matches)
    return dev


# stereo-check kinds (NVMK_CHECK_* of include/nvmolkit_amd.h)
CHECK_TETRAHEDRAL, CHECK_CHIRAL_VOLUME, CHECK_CHIRAL_DISTANCE, CHECK_CHIRAL_CENTER_VOLUME, CHECK_DOUBLE_BOND_STEREO, \
    CHECK_DOUBLE_BOND_GEOMETRY = range(6)


def stereo_check_flat(kind: int, positions: torch.Tensor, atom_starts, sys_mol, check_starts, check_kind, check_idx, check_par,
                      active=None, stream=None) -> torch.Tensor:
    """One stereochemistry check stage (``CHECK_*`` kind) on given coordinates — the unit the embedding pipeline runs
    between its minimisations (reference: src/etkdg_stage_stereochem_checks.cu).  ``positions`` is a float64 CUDA tensor
    (total_atoms, 4) (x, y, z, w), ``atom_starts`` has n_systems + 1 entries, ``sys_mol[s]`` selects the molecule whose
    check terms (``check_starts`` per molecule; ``check_kind``, ``check_idx`` (n, 5), ``check_par`` (n, 2)) apply to
    system s.  Returns a uint8 CUDA tensor (n_systems,): 1 where a term of that kind failed."""
    if not (isinstance(positions, torch.Tensor) and positions.is_cuda and positions.dtype == torch.float64 and positions.dim() == 2
            and positions.shape[1] == 4):
        raise ValueError("positions must be a float64 CUDA tensor of shape (total_atoms, 4)")
    dev = positions.device
    to = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt)).to(dev)  # noqa: E731
    d_as, d_sm = to(atom_starts, np.int32), to(sys_mol, np.int32)
    d_cs, d_ck = to(check_starts, np.int32), to(check_kind, np.int32)
    d_ci, d_cp = to(np.asarray(check_idx).reshape(-1, 5), np.int32), to(np.asarray(check_par).reshape(-1, 2), np.float64)
    n_sys = int(d_sm.numel())
    if int(d_as.numel()) != n_sys + 1:
        raise ValueError("atom_starts must have n_systems + 1 entries")
    failed = torch.zeros(n_sys, dtype=torch.uint8, device=dev)
    d_act = None if active is None else torch.as_tensor(active).to(device=dev, dtype=torch.uint8).contiguous()
    pos = positions.contiguous()
    with torch.cuda.device(dev):
        rc = _native.lib().nvmk_etkdg_stereo_check(int(kind), n_sys, d_as.data_ptr(), d_sm.data_ptr(), d_cs.data_ptr(), d_ck.data_ptr(),
                                                   d_ci.data_ptr(), d_cp.data_ptr(), pos.data_ptr(),
                                                   0 if d_act is None else d_act.data_ptr(), failed.data_ptr(),
                                                   _native.stream_ptr(stream))
    _native.check(rc, "nvmk_etkdg_stereo_check")
    return failed


def embed_flat_molecules(flat_mols: Sequence[FlatMolecule], confs_per_molecule: int, max_iterations: int,
                         hardware_options: HardwareOptions | None, output: CoordinateOutput, target_gpu: int | None, **kw):
    """ETKDG on flattened molecules over the GPUs of ``hardware_options.gpuIds``: molecules are dealt to the GPUs largest
    first (cost ~ atoms^2, like the reference's sort in src/etkdg.cpp:151-154) and every GPU embeds its share on a host
    thread and stream of its own (src/etkdg.cpp:211-240); no data moves between GPUs until the results are gathered.
    Returns a list of (k_m, n_atoms_m, 3) host arrays in input order, or with ``output=DEVICE`` one
    :class:`Device3DResult` on ``target_gpu`` whose ``mol_indices`` are input positions."""
    from nvmolkit_amd._rdkit_confs import run_per_gpu

    opts = hardware_options or HardwareOptions()
    gpu_ids = list(opts.gpuIds) if opts.gpuIds else [torch.cuda.current_device()]
    if output == CoordinateOutput.DEVICE:
        if target_gpu is None or target_gpu < 0:
            target_gpu = gpu_ids[0]
        if target_gpu not in gpu_ids:
            raise ValueError(f"targetGpu {target_gpu} is not in the configured set of execution GPUs; pass it via "
                             "hardwareOptions.gpuIds first.")
    n_atoms = np.array([m.n_atoms for m in flat_mols], dtype=np.int64)
    order = np.argsort(-n_atoms, kind="stable")
    shares = [order[slot::len(gpu_ids)] for slot in range(len(gpu_ids))]
    results: list = [None] * len(gpu_ids)

    def gpu_worker(slot: int) -> None:
        mine = shares[slot]
        if len(mine) == 0:
            return
        device = torch.device("cuda", gpu_ids[slot])
        with torch.cuda.device(device):
            stream = torch.cuda.Stream(device=device)
            with torch.cuda.stream(stream):
                # assembled by opts.preprocessingThreads host threads and uploaded on this stream; the embedding is queued behind it
                # (an unset thread count is this PROCESS's share of the cores: the GPUs' builders run side by side — ADVICE r05)
                threads = opts.preprocessingThreads if opts.preprocessingThreads > 0 else max(1, _native.build_threads(-1) // len(gpu_ids))
                molset = FlatMoleculeSet([flat_mols[i] for i in mine], device=device, preprocessing_threads=threads)
                results[slot] = embed_flat(molset, confs_per_molecule, max_iterations,
                                           batch_size=opts.batchSize if opts.batchSize > 0 else -1,
                                           batches_per_gpu=opts.batchesPerGpu, stream=stream, output=output, **kw)
            stream.synchronize()

    run_per_gpu(gpu_worker, len(gpu_ids))
    if output != CoordinateOutput.DEVICE:
        per_mol: list = [np.zeros((0, int(n), 3)) for n in n_atoms]
        for slot, res in enumerate(results):
            if res is None:
                continue
            for local, m in enumerate(shares[slot]):
                per_mol[int(m)] = res.conformers(local).cpu().numpy()
        return per_mol
    # consolidate on the target GPU, conformers ordered by input molecule (detail::finalizeOnTarget)
    tgt = torch.device("cuda", target_gpu)
    parts = []
    for slot, res in enumerate(results):
        if res is None:
            continue
        g = torch.from_numpy(np.asarray(shares[slot], dtype=np.int64)).to(tgt)
        parts.append((res.values.torch().to(tgt), res.atom_starts.torch().to(tgt).to(torch.int64),
                      g[res.mol_indices.torch().to(tgt).to(torch.int64)], res.conf_indices.torch().to(tgt).to(torch.int64)))
    if not parts:
        z = lambda dt: torch.zeros(0, dtype=dt, device=tgt)  # noqa: E731
        return Device3DResult(torch.zeros((0, 3), dtype=torch.float64, device=tgt), torch.zeros(1, dtype=torch.int32, device=tgt),
                              z(torch.int32), z(torch.int32), target_gpu, len(flat_mols))
    sizes = torch.cat([p[1][1:] - p[1][:-1] for p in parts])
    mols = torch.cat([p[2] for p in parts])
    confs = torch.cat([p[3] for p in parts])
    row0 = torch.cat([p[1][:-1] + off for p, off in zip(parts, np.cumsum([0] + [int(p[0].shape[0]) for p in parts[:-1]]))])
    values = torch.cat([p[0] for p in parts])
    key = torch.argsort(mols * (int(confs.max().item()) + 1 if confs.numel() else 1) + confs, stable=True)
    sizes, mols, confs, row0 = sizes[key], mols[key], confs[key], row0[key]
    starts = torch.zeros(sizes.numel() + 1, dtype=torch.int64, device=tgt)
    starts[1:] = torch.cumsum(sizes, 0)
    rows = torch.repeat_interleave(row0 - starts[:-1], sizes) + torch.arange(int(starts[-1].item()), device=tgt)
    return Device3DResult(values[rows], starts.to(torch.int32), mols.to(torch.int32), confs.to(torch.int32), target_gpu,
                          len(flat_mols))


def EmbedMolecules(molecules, params, confsPerMolecule: int = 1, maxIterations: int = -1,
                   hardwareOptions: HardwareOptions | None = None, output: CoordinateOutput = CoordinateOutput.RDKIT_CONFORMERS,
                   targetGpu: int = -1):
    """Embed multiple molecules with multiple conformers on the GPU(s) (reference: nvmolkit/embedMolecules.py:55-158).

    Chemistry perception is RDKit's through its public Python API (``nvmolkit_amd._rdkit_embed``: smoothed bounds matrix,
    experimental torsions, chiral sets, double-bond lists -> flattened term tables; the reference does the same in C++,
    src/embedder_utils.cpp:662-712).  ``RDKIT_CONFORMERS`` writes the conformers into the molecules (existing conformers
    are cleared, ids 0..k-1) and returns ``None``; ``DEVICE`` leaves the molecules untouched and returns a
    :class:`Device3DResult` on ``targetGpu`` (default: the first execution GPU).  Restrictions as the reference:
    ``useRandomCoords`` must be True; bounds matrices, coordinate maps and custom CPCI are not supported; all fragments are
    embedded together.  ``params.pruneRmsThresh > 0`` prunes the conformers on the GPU (RDKIT_CONFORMERS only, like the
    reference)."""
    if confsPerMolecule <= 0:
        raise ValueError("confsPerMolecule must be greater than 0")
    if maxIterations < -1 or maxIterations == 0:
        raise ValueError("maxIterations must be -1 (automatic) or greater than 0")
    if not molecules:
        if output == CoordinateOutput.DEVICE:
            raise ValueError("EmbedMolecules(output=DEVICE) requires at least one molecule")
        return None
    for i, mol in enumerate(molecules):
        if mol is None:
            raise ValueError(f"Molecule at index {i} is None")
    if not getattr(params, "useRandomCoords", False):
        raise ValueError("ETKDG requires useRandomCoords=True in EmbedParameters")  # nvmolkit/embedMolecules.py:146-147
    prune = float(getattr(params, "pruneRmsThresh", -1.0))
    if output is CoordinateOutput.DEVICE and prune > 0:
        raise ValueError("RMS pruning is not supported with DEVICE output")
    for name in ("boundsMat", "coordMap", "CPCI"):
        if getattr(params, name, None):
            raise NotImplementedError(f"EmbedParameters.{name} is not supported (as in the reference)")
    from nvmolkit_amd import _rdkit_embed

    flat = [FlatMolecule(**_rdkit_embed.flatten_etkdg_from_rdkit(m, params)) for m in molecules]
    seed = int(getattr(params, "randomSeed", -1))
    if seed < 0:
        seed = int(np.random.SeedSequence().entropy & 0x7FFFFFFFFFFFFFFF)
    kw = dict(use_exp_torsions=bool(getattr(params, "useExpTorsionAnglePrefs", False)),
              use_basic_knowledge=bool(getattr(params, "useBasicKnowledge", False)),
              enforce_chirality=bool(getattr(params, "enforceChirality", True)),
              box_size_mult=float(getattr(params, "boxSizeMult", 2.0)),
              force_tol=float(getattr(params, "optimizerForceTol", 1e-3)), seed=seed)
    if output is CoordinateOutput.DEVICE:
        return embed_flat_molecules(flat, confsPerMolecule, maxIterations, hardwareOptions, output, targetGpu, **kw)
    coords = embed_flat_molecules(flat, confsPerMolecule, maxIterations, hardwareOptions, output, None, **kw)
    if prune > 0:
        coords = [_prune_host(c, prune, m, params) for c, m in zip(coords, molecules)]
    for mol, xyz in zip(molecules, coords):
        _rdkit_embed.write_conformers(mol, xyz)
    return None


def _prune_host(xyz: np.ndarray, threshold: float, mol, params) -> np.ndarray:
    """RMS pruning of one molecule's conformers with the GPU pruning kernels (conformerRmsd.prune_conformers), on the atoms and
    with the self matches the reference uses (getMolSelfMatches, rdkit_extensions/conformer_pruning.cpp:24-72)."""
    if len(xyz) < 2:
        return xyz
    from nvmolkit_amd.conformerRmsd import prune_conformers

    n = xyz.shape[1]
    dev = torch.device("cuda", torch.cuda.current_device())
    starts = torch.arange(0, (len(xyz) + 1) * n, n, dtype=torch.int32, device=dev)
    res = Device3DResult(torch.from_numpy(np.ascontiguousarray(xyz.reshape(-1, 3))).to(dev), starts,
                         torch.zeros(len(xyz), dtype=torch.int32, device=dev),
                         torch.arange(len(xyz), dtype=torch.int32, device=dev), dev.index, 1)
    subset = matches = None
    if bool(getattr(params, "useSymmetryForPruning", False)):
        from nvmolkit_amd import _rdkit_embed

        matches = [_rdkit_embed.self_matches_for_pruning(mol, bool(getattr(params, "symmetrizeConjugatedTerminalGroupsForPruning", True)))]
    elif bool(getattr(params, "onlyHeavyAtomsForRMS", False)):
        subset = [[a.GetIdx() for a in mol.GetAtoms() if a.GetAtomicNum() > 1]]
    kept = prune_conformers(res, threshold, subset, matches)
    return kept.values.torch().cpu().numpy().reshape(-1, n, 3)


def random_coords_flat(seed: int, attempt_base: int, atom_starts, box_size: float, active=None, device="cuda", stream=None):
    """Stage 0 of the pipeline on its own: uniform 4-D start coordinates in [-box_size / 2, box_size / 2) for every active
    system; system s is attempt ``attempt_base + s`` of the seeded run (reference: ETKDGCoordGenStage,
    src/etkdg_stage_coordgen.cu:83-127).  Returns a float64 CUDA tensor (total_atoms, 4); inactive systems stay zero."""
    dev = torch.device(device)
    d_as = torch.as_tensor(np.ascontiguousarray(atom_starts, dtype=np.int32)).to(dev)
    n_sys = int(d_as.numel()) - 1
    pos = torch.zeros((int(np.asarray(atom_starts)[-1]), 4), dtype=torch.float64, device=dev)
    d_act = None if active is None else torch.as_tensor(active).to(device=dev, dtype=torch.uint8).contiguous()
    with torch.cuda.device(dev):
        rc = _native.lib().nvmk_etkdg_random_coords(int(seed) & 0xFFFFFFFFFFFFFFFF, int(attempt_base), n_sys, d_as.data_ptr(),
                                                    0 if d_act is None else d_act.data_ptr(), float(box_size), pos.data_ptr(),
                                                    _native.stream_ptr(stream))
    _native.check(rc, "nvmk_etkdg_random_coords")
    return pos


def driver_run_programmed(failed, max_iterations: int, stream=None):
    """The ETKDG driver's bookkeeping with programmed stages (reference: ETKDGDriver with ProgrammableStep stages,
    tests/test_etkdg.cu:41-341).  ``failed[stage][iteration][system]`` (0 / 1); returns ``(fail_counts (n_stages,
    n_systems), finished_on (n_systems,), n_finished, iterations_complete)``."""
    f = np.ascontiguousarray(failed, dtype=np.uint8)
    if f.ndim != 3:
        raise ValueError("failed must be (n_stages, n_iterations, n_systems)")
    n_stages, n_it, n_sys = f.shape
    if n_it < max_iterations:
        f = np.concatenate([f, np.zeros((n_stages, max_iterations - n_it, n_sys), dtype=np.uint8)], axis=1)
    f = np.ascontiguousarray(f[:, :max_iterations])
    counts = np.zeros((max(n_stages, 1), max(n_sys, 1)), dtype=np.int16)
    fin = np.full(max(n_sys, 1), -1, dtype=np.int16)
    nf, it = ctypes.c_int32(0), ctypes.c_int32(0)
    rc = _native.lib().nvmk_etkdg_driver_run(n_sys, n_stages, int(max_iterations), f.ctypes.data, counts.ctypes.data, fin.ctypes.data,
                                             ctypes.byref(nf), ctypes.byref(it), _native.stream_ptr(stream))
    _native.check(rc, "nvmk_etkdg_driver_run")
    return counts[:n_stages, :n_sys], fin[:n_sys], nf.value, it.value
