"""What the library's table builder (nvmolkit_amd/csrc/table_build.cpp: nvmk_etkdg_molset_build, nvmk_ff_tables_build) must
produce, restated with torch / numpy on the CPU: stacking of per-molecule term groups, the diagonal order of the pair tables
and the merged MMFF non-bonded table.  The two torch functions are the product's table preparation of rounds 2-4 (then run on
the device, nvmolkit_amd/forcefield.py) moved here unchanged: the native builder is held to them row by row.  Also readers
that turn the builder's views (raw host or device pointers) back into numpy arrays."""

from __future__ import annotations

import ctypes

import numpy as np
import torch

from nvmolkit_amd import _native
from nvmolkit_amd.forcefield import CONSTRAINT_LAYOUT, DG, ETK, GROUP_LAYOUT, MMFF, PAIR_ORDER_GROUPS, stack_molecule_tables


def diagonal_pair_order(starts: torch.Tensor, idx: torch.Tensor, par: torch.Tensor):
    """Rows of a pair-term group inside every system ordered by (|j - i|, min(i, j)), stable."""
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


def expected_term_tables(kind: int, tables, pair_order: bool = True, merge: bool = True):
    """``tables[m][g] = (idx, par)`` -> list of (starts, idx, par) numpy groups as the builder lays them out, and the merged
    MMFF group (or None)."""
    return expected_from_stacked(kind, stack_molecule_tables(kind, tables), pair_order, merge)


def expected_from_stacked(kind: int, stacked, pair_order: bool = True, merge: bool = True):
    groups = []
    for g, (starts, idx, par) in enumerate(stacked):
        t = (torch.from_numpy(np.ascontiguousarray(starts, dtype=np.int32)), torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int32)),
             torch.from_numpy(np.ascontiguousarray(par, dtype=np.float64)))
        if pair_order and g in PAIR_ORDER_GROUPS[kind]:
            t = (t[0],) + tuple(diagonal_pair_order(*t))
        groups.append(t)
    merged = None
    if kind == MMFF and merge:
        raw = [(torch.from_numpy(np.ascontiguousarray(s, dtype=np.int32)), torch.from_numpy(np.ascontiguousarray(i, dtype=np.int32)),
                torch.from_numpy(np.ascontiguousarray(p, dtype=np.float64))) for s, i, p in (stacked[5], stacked[6])]
        merged = merge_mmff_nonbonded(*raw)
        if merged is not None and pair_order:
            merged = (merged[0],) + tuple(diagonal_pair_order(*merged))
    as_np = lambda t: tuple(x.numpy() for x in t)  # noqa: E731
    return [as_np(t) for t in groups], (as_np(merged) if merged is not None else None)


def expected_molset(mols, pair_order: bool = True):
    """FlatMolecule list -> dict of what nvmk_etkdg_molset_view must show (the FlatMoleculeSet.__init__ of rounds 1-4)."""
    out = {"n_atoms": np.array([m.n_atoms for m in mols], dtype=np.int32),
           "num_impropers": np.array([m.num_impropers for m in mols], dtype=np.int32)}
    out["dg"], _ = expected_term_tables(DG, [m.dg for m in mols], pair_order)
    has_etk = bool(mols) and all(m.etk is not None for m in mols)
    out["etk"] = expected_term_tables(ETK, [m.etk for m in mols], pair_order)[0] if has_etk else None
    starts = np.zeros(len(mols) + 1, dtype=np.int32)
    kinds, idxs, pars = [], [], []
    for i, m in enumerate(mols):
        starts[i + 1] = starts[i] + len(m.checks)
        for kind, idx, par in m.checks:
            kinds.append(kind)
            idxs.append(list(idx) + [0] * (5 - len(idx)))
            pars.append(list(par) + [0.0] * (2 - len(par)))
    out["checks"] = (starts, np.array(kinds, dtype=np.int32), np.array(idxs, dtype=np.int32).reshape(-1, 5),
                     np.array(pars, dtype=np.float64).reshape(-1, 2)) if kinds else None
    return out


# ---- reading the builder's views ------------------------------------------------------------------------------------------
_hip = None


def _read(ptr, nbytes: int, on_device: bool) -> bytes:
    if not ptr or nbytes == 0:
        return b""
    if not on_device:
        return ctypes.string_at(ptr, nbytes)
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")  # the copy torch already loaded
        _hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    buf = ctypes.create_string_buffer(nbytes)
    torch.cuda.synchronize()
    assert _hip.hipMemcpy(buf, ptr, nbytes, 2) == 0, "hipMemcpy device -> host failed"
    return buf.raw


def read_group(group, n_rows: int, n_idx: int, n_par: int, on_device: bool):
    """One nvmk_ff_group (``_native.FFGroup``) -> (starts, idx, par) numpy arrays; None when the group is absent."""
    if not group.starts:
        return None
    starts = np.frombuffer(_read(group.starts, 4 * (n_rows + 1), on_device), dtype=np.int32)
    n = int(starts[-1])
    idx = np.frombuffer(_read(group.idx, 4 * n * n_idx, on_device), dtype=np.int32).reshape(n, n_idx)
    par = np.frombuffer(_read(group.par, 8 * n * n_par, on_device), dtype=np.float64).reshape(n, n_par)
    return starts, idx, par


def read_tables(tables):
    """MoleculeTermTables -> (groups, merged) like :func:`expected_term_tables`."""
    on_device = tables.device.type == "cuda"
    layout = list(GROUP_LAYOUT[tables.kind]) + list(CONSTRAINT_LAYOUT)
    groups = []
    for g in range(11):
        t = read_group(tables.view[g], tables.n_mols, *layout[g], on_device) if g < len(layout) else None
        if t is None:
            break
        groups.append(t)
    return groups, read_group(tables.view[11], tables.n_mols, 2, 5, on_device)


def read_molset(molset):
    """FlatMoleculeSet -> dict like :func:`expected_molset`."""
    c, n = molset.c, molset.c.n_mols
    on_device = molset.device.type == "cuda"
    if on_device:  # (an asynchronous build: every row uploaded before the block is read back)
        import torch

        molset.wait()
        torch.cuda.synchronize()
    out = {"n_atoms": np.frombuffer(ctypes.string_at(c.h_n_atoms, 4 * n), dtype=np.int32) if n else np.zeros(0, np.int32),
           "num_impropers": np.frombuffer(_read(c.num_impropers, 4 * n, on_device), dtype=np.int32)}
    out["dg"] = [read_group(c.dg[g], n, *GROUP_LAYOUT[DG][g], on_device) for g in range(3)]
    out["etk"] = [read_group(c.etk[g], n, *GROUP_LAYOUT[ETK][g], on_device) for g in range(6)] if c.h_etk_d12_counts else None
    if c.check_starts:
        starts = np.frombuffer(_read(c.check_starts, 4 * (n + 1), on_device), dtype=np.int32)
        k = int(starts[-1])
        out["checks"] = (starts, np.frombuffer(_read(c.check_kind, 4 * k, on_device), dtype=np.int32),
                         np.frombuffer(_read(c.check_idx, 20 * k, on_device), dtype=np.int32).reshape(k, 5),
                         np.frombuffer(_read(c.check_par, 16 * k, on_device), dtype=np.float64).reshape(k, 2))
    else:
        out["checks"] = None
    if c.h_etk_d12_counts:
        out["d12"] = np.frombuffer(ctypes.string_at(c.h_etk_d12_counts, 4 * n), dtype=np.int32)
        out["d13"] = np.frombuffer(ctypes.string_at(c.h_etk_d13_counts, 4 * n), dtype=np.int32)
    return out


def assert_groups_equal(got, want, what=""):
    assert (got is None) == (want is None), f"{what}: present / absent mismatch"
    if got is None:
        return
    for name, a, b in zip(("starts", "idx", "par"), got, want):
        assert a.shape == b.shape and np.array_equal(a, b), f"{what}: {name} differs"
