"""Pairwise conformer RMSD matrices and RMS pruning on the GPU (reference API: nvmolkit/conformerRmsd.py:30-156;
kernels src/conformer_rmsd.cu; pruning rdkit_extensions/conformer_pruning.cpp:88-137).

``GetConformerRMSMatrix`` / ``GetConformerRMSMatrixBatch`` keep the reference's names, arguments, condensed
lower-triangle result order (pair (i, j), i > j, at ``i*(i-1)//2 + j``) and error behaviour; they only read conformer
positions, so any object with ``GetNumAtoms()`` / ``GetConformers()`` / ``conf.GetPositions()`` works.  The
``*_flat`` functions take coordinate tensors and are what the rest of the package (pruning of ETKDG output) uses.
"""

from __future__ import annotations

import numpy as np
import torch

from nvmolkit_amd import _native
from nvmolkit_amd.types import AsyncGpuResult, Device3DResult

__all__ = ["GetConformerRMSMatrix", "GetConformerRMSMatrixBatch", "conformer_rms_matrix_flat", "conformer_rms_matrix_sym_flat",
           "prune_conformers"]


def _check_stream(stream):
    if stream is not None and not isinstance(stream, torch.cuda.Stream):
        raise TypeError(f"stream must be a torch.cuda.Stream or None, got {type(stream).__name__}")


def conformer_rms_matrix_flat(coords: list[torch.Tensor], prealigned: bool = False, stream=None) -> list[torch.Tensor]:
    """RMSD matrices of a batch: ``coords[m]`` is a float64 CUDA tensor (n_confs_m, n_atoms_m, 3); returns one 1-D tensor of
    n (n - 1) / 2 values per molecule (views of one buffer), all computed by ONE launch."""
    _check_stream(stream)
    if not coords:
        return []
    device = coords[0].device
    n_confs = np.array([int(c.shape[0]) for c in coords], dtype=np.int64)
    n_atoms = np.array([int(c.shape[1]) if c.dim() == 3 else 0 for c in coords], dtype=np.int64)
    for m, c in enumerate(coords):
        if not (isinstance(c, torch.Tensor) and c.is_cuda and c.dtype == torch.float64 and c.dim() == 3 and c.shape[2] == 3):
            raise ValueError(f"coords[{m}] must be a float64 CUDA tensor of shape (n_confs, n_atoms, 3)")
        if n_confs[m] > 0 and n_atoms[m] == 0:
            raise ValueError(f"molecule {m} has conformers but no atoms")
    pairs = n_confs * (n_confs - 1) // 2
    pair_off = np.zeros(len(coords) + 1, dtype=np.int64)
    pair_off[1:] = np.cumsum(pairs)
    coord_off = np.zeros(len(coords) + 1, dtype=np.int64)
    coord_off[1:] = np.cumsum(n_confs * n_atoms * 3)
    total = int(pair_off[-1])
    with _native.on_stream(stream, device):  # staging tensors and the kernel share one stream
        out = torch.empty(total, dtype=torch.float64, device=device)
        if total > 0:
            flat = torch.cat([c.reshape(-1) for c in coords]) if len(coords) > 1 else coords[0].reshape(-1).contiguous()
            to_dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(device)  # noqa: E731
            d_coff, d_na, d_poff = to_dev(coord_off, np.int64), to_dev(n_atoms, np.int32), to_dev(pair_off, np.int64)
            with torch.cuda.device(device):
                rc = _native.lib().nvmk_conformer_rmsd_batch(flat.data_ptr(), d_coff.data_ptr(), d_na.data_ptr(),
                                                             d_poff.data_ptr(), len(coords), total, int(bool(prealigned)),
                                                             out.data_ptr(), _native.stream_ptr(stream))
            _native.check(rc, "nvmk_conformer_rmsd_batch")
    return [out[pair_off[m]:pair_off[m + 1]] for m in range(len(coords))]


def conformer_rms_matrix_sym_flat(coords: list[torch.Tensor], matches: list, stream=None) -> list[torch.Tensor]:
    """Symmetry-aware RMSD matrices: ``matches[m]`` is an integer array (K_m, L_m) of atom mappings of molecule m (row 0 = the
    reference atoms, normally the identity over the heavy atoms; rows k > 0 the molecule's other self matches — RDKit's
    ``mol.GetSubstructMatches(mol, uniquify=False, maxMatches=1000)`` on the hydrogen-stripped molecule, or
    ``SmilesSet.self_matches``).  Entry (i, j) is the smallest optimally superposed RMSD of conformer i's atoms ``matches[m][0]``
    against conformer j's atoms ``matches[m][k]`` over k — the quantity the reference's ``_isConfFarFromRest`` thresholds
    (rdkit_extensions/conformer_pruning.cpp:88-114).  One launch for the batch."""
    _check_stream(stream)
    if not coords:
        return []
    if len(matches) != len(coords):
        raise ValueError("matches must have one entry per molecule")
    device = coords[0].device
    n_confs = np.array([int(c.shape[0]) for c in coords], dtype=np.int64)
    n_atoms = np.array([int(c.shape[1]) for c in coords], dtype=np.int64)
    tables = []
    for m, (c, mt) in enumerate(zip(coords, matches)):
        if not (isinstance(c, torch.Tensor) and c.is_cuda and c.dtype == torch.float64 and c.dim() == 3 and c.shape[2] == 3):
            raise ValueError(f"coords[{m}] must be a float64 CUDA tensor of shape (n_confs, n_atoms, 3)")
        t = np.ascontiguousarray(np.asarray(mt, dtype=np.int32))
        if t.ndim != 2 or t.shape[0] < 1 or t.shape[1] < 1:
            raise ValueError(f"matches[{m}] must be a non-empty (K, L) integer array")
        if t.min() < 0 or t.max() >= n_atoms[m]:
            raise ValueError(f"matches[{m}] names atoms outside the molecule")
        tables.append(t)
    pairs = n_confs * (n_confs - 1) // 2
    pair_off = np.zeros(len(coords) + 1, dtype=np.int64)
    pair_off[1:] = np.cumsum(pairs)
    coord_off = np.zeros(len(coords) + 1, dtype=np.int64)
    coord_off[1:] = np.cumsum(n_confs * n_atoms * 3)
    match_off = np.zeros(len(coords) + 1, dtype=np.int64)
    match_off[1:] = np.cumsum([t.size for t in tables])
    match_len = np.array([t.shape[1] for t in tables], dtype=np.int32)
    total = int(pair_off[-1])
    with _native.on_stream(stream, device):
        out = torch.empty(total, dtype=torch.float64, device=device)
        if total > 0:
            flat = torch.cat([c.reshape(-1) for c in coords]) if len(coords) > 1 else coords[0].reshape(-1).contiguous()
            to_dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(device)  # noqa: E731
            d_coff, d_na, d_poff = to_dev(coord_off, np.int64), to_dev(n_atoms, np.int32), to_dev(pair_off, np.int64)
            d_moff, d_mlen, d_m = to_dev(match_off, np.int64), to_dev(match_len, np.int32), to_dev(np.concatenate([t.reshape(-1) for t in tables]), np.int32)
            with torch.cuda.device(device):
                rc = _native.lib().nvmk_conformer_rmsd_batch_sym(flat.data_ptr(), d_coff.data_ptr(), d_na.data_ptr(), d_poff.data_ptr(),
                                                                 len(coords), total, d_moff.data_ptr(), d_mlen.data_ptr(), d_m.data_ptr(),
                                                                 out.data_ptr(), _native.stream_ptr(stream))
            _native.check(rc, "nvmk_conformer_rmsd_batch_sym")
    return [out[pair_off[m]:pair_off[m + 1]] for m in range(len(coords))]


def _positions(mol, device) -> torch.Tensor:
    confs = list(mol.GetConformers())
    n = mol.GetNumAtoms()
    if confs and n == 0:
        raise ValueError("molecule has conformers but no atoms")
    xyz = np.stack([np.asarray(c.GetPositions(), dtype=np.float64).reshape(n, 3) for c in confs]) if confs else np.zeros((0, n, 3))
    return torch.from_numpy(xyz).to(device)


def GetConformerRMSMatrix(mol, prealigned: bool = False, stream=None) -> AsyncGpuResult:
    """GPU equivalent of ``AllChem.GetConformerRMSMatrix(mol, prealigned=prealigned)``: N (N - 1) / 2 pairwise RMSDs over all
    atoms of ``mol`` (strip hydrogens first for heavy-atom RMSD), each pair optimally superimposed unless ``prealigned``."""
    if mol is None:
        raise ValueError("mol must not be None")
    _check_stream(stream)
    return AsyncGpuResult(conformer_rms_matrix_flat([_positions(mol, torch.device("cuda", torch.cuda.current_device()))],
                                                    prealigned, stream)[0])


def GetConformerRMSMatrixBatch(mols, prealigned: bool = False, stream=None) -> list[AsyncGpuResult]:
    """One launch for a list of molecules; molecules with fewer than 2 conformers return an empty result."""
    _check_stream(stream)
    for i, mol in enumerate(mols):
        if mol is None:
            raise ValueError(f"mol at index {i} must not be None")
    device = torch.device("cuda", torch.cuda.current_device())
    return [AsyncGpuResult(t) for t in conformer_rms_matrix_flat([_positions(m, device) for m in mols], prealigned, stream)]


def prune_conformers(conformers: Device3DResult, threshold: float, atom_subsets=None, self_matches=None) -> Device3DResult:
    """RMS pruning of a :class:`Device3DResult` on its GPU (EmbedParameters.pruneRmsThresh; reference:
    addConformersToMoleculeWithPruning): per molecule, in conformer order, a conformer is kept iff its aligned RMSD to every
    conformer kept before it is >= ``threshold``.  ``atom_subsets[m]`` (optional index array) restricts the RMSD of molecule
    m to those atoms (``onlyHeavyAtomsForRMS``).  ``self_matches[m]`` (optional (K, L) integer array, row 0 the reference atoms)
    makes the pruning symmetry-aware as in the reference (``useSymmetryForPruning``: getMolSelfMatches /
    _isConfFarFromRest, rdkit_extensions/conformer_pruning.cpp:24-114): the RMSD of a pair is the smallest over the molecule's
    self matches, so conformers that differ by a permutation of equivalent atoms count as the same.  The matches come from
    RDKit where it exists (``mol.GetSubstructMatches(mol, uniquify=False, maxMatches=1000)`` on the hydrogen-stripped molecule)
    or from the library's own ingestion (``SmilesSet.self_matches``).  Returns a compacted result (conformer indices
    renumbered 0..k-1 per molecule)."""
    if atom_subsets is not None and self_matches is not None:
        raise ValueError("pass atom_subsets or self_matches (whose rows already name the atoms), not both")
    if threshold <= 0.0:
        return conformers
    values = conformers.values.torch()
    dev = values.device
    starts = conformers.atom_starts.torch().to(torch.int64)
    mols = conformers.mol_indices.torch().to(torch.int64)
    n_conf = mols.numel()
    if n_conf == 0:
        return conformers
    starts_h, mols_h = starts.cpu().numpy(), mols.cpu().numpy()
    if np.any(np.diff(mols_h) < 0):
        raise ValueError("conformers must be grouped by molecule")
    conf_starts = np.searchsorted(mols_h, np.arange(conformers.n_mols + 1)).astype(np.int32)  # conformers of molecule m
    coords = []
    for m in range(conformers.n_mols):
        c0, c1 = int(conf_starts[m]), int(conf_starts[m + 1])
        if c1 == c0:
            coords.append(torch.zeros((0, 1, 3), dtype=torch.float64, device=dev))
            continue
        n = int(starts_h[c0 + 1] - starts_h[c0])
        block = values[int(starts_h[c0]):int(starts_h[c1])].view(c1 - c0, n, 3)
        if atom_subsets is not None and atom_subsets[m] is not None:
            block = block[:, torch.as_tensor(atom_subsets[m], dtype=torch.int64, device=dev)]
        coords.append(block.contiguous())
    if self_matches is not None:
        tables = [np.asarray(self_matches[m], dtype=np.int32) if self_matches[m] is not None and coords[m].shape[0] > 0
                  else np.arange(max(int(coords[m].shape[1]), 1), dtype=np.int32)[None, :] for m in range(conformers.n_mols)]
        mats = conformer_rms_matrix_sym_flat(coords, tables)
    else:
        mats = conformer_rms_matrix_flat(coords)
    counts = np.diff(conf_starts).astype(np.int64)
    pair_off = np.zeros(conformers.n_mols + 1, dtype=np.int64)
    pair_off[1:] = np.cumsum(counts * (counts - 1) // 2)
    rmsd = torch.cat(mats) if mats else torch.zeros(0, dtype=torch.float64, device=dev)
    keep = torch.zeros(n_conf, dtype=torch.uint8, device=dev)
    d_poff = torch.from_numpy(pair_off).to(dev)
    d_cs = torch.from_numpy(conf_starts).to(dev)
    with torch.cuda.device(dev):
        rc = _native.lib().nvmk_conformer_prune(rmsd.data_ptr() if rmsd.numel() else None, d_poff.data_ptr(), d_cs.data_ptr(),
                                                conformers.n_mols, float(threshold), keep.data_ptr(), _native.stream_ptr(None))
    _native.check(rc, "nvmk_conformer_prune")
    kept = keep.bool()
    sizes = starts[1:] - starts[:-1]
    row_keep = torch.repeat_interleave(kept, sizes)
    new_sizes = sizes[kept]
    new_starts = torch.zeros(new_sizes.numel() + 1, dtype=torch.int64, device=dev)
    new_starts[1:] = torch.cumsum(new_sizes, 0)
    new_mols = mols[kept]
    first = torch.searchsorted(new_mols, new_mols)  # index of the first kept conformer of the same molecule
    new_conf = torch.arange(new_mols.numel(), device=dev) - first
    pick = lambda t: None if t is None else t.torch()[kept]  # noqa: E731
    return Device3DResult(values[row_keep], new_starts.to(torch.int32), new_mols.to(torch.int32), new_conf.to(torch.int32),
                          conformers.gpu_id, conformers.n_mols, energies=pick(conformers.energies),
                          converged=pick(conformers.converged))
