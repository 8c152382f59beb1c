"""ctypes front end of oracle/oracle_ff.c — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The C file is the fast twin of oracle/ff.py: analytic gradients, RDKit's BFGS and the ETKDG stage pipeline on the same
flattened term tables the product consumes, OpenMP over systems.  It is the CPU baseline of bench.py's conformer block and
the oracle of the large-system BFGS / ETKDG parity tests; tests/test_oracle_ff_c.py pins it against oracle/ff.py first.
"""

from __future__ import annotations

import ctypes

import numpy as np

import oracle
from oracle import ff as off

N_STAGES = 11


class _Group(ctypes.Structure):
    _fields_ = [("starts", ctypes.c_void_p), ("idx", ctypes.c_void_p), ("par", ctypes.c_void_p)]


class _Batch(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("n_systems", ctypes.c_int32), ("atom_starts", ctypes.c_void_p),
                ("groups", _Group * 12), ("system_mol", ctypes.c_void_p), ("group_mask", ctypes.c_uint32),
                ("etk_ref12_starts", ctypes.c_void_p), ("etk_ref12", ctypes.c_void_p),
                ("etk_ref13_starts", ctypes.c_void_p), ("etk_ref13", ctypes.c_void_p)]


class _Molset(ctypes.Structure):
    _fields_ = [("n_mols", ctypes.c_int32), ("n_atoms", ctypes.c_void_p), ("dg", _Group * 3), ("etk", _Group * 6),
                ("check_starts", ctypes.c_void_p), ("check_kind", ctypes.c_void_p), ("check_idx", ctypes.c_void_p),
                ("check_par", ctypes.c_void_p), ("num_impropers", ctypes.c_void_p)]


class _Params(ctypes.Structure):
    _fields_ = [("confs_per_mol", ctypes.c_int32), ("max_iterations", ctypes.c_int32), ("batch_size", ctypes.c_int32),
                ("use_exp_torsions", ctypes.c_int32), ("use_basic_knowledge", ctypes.c_int32),
                ("enforce_chirality", ctypes.c_int32), ("box_size", ctypes.c_double), ("force_tol", ctypes.c_double),
                ("seed", ctypes.c_uint64)]


_declared = False


def _lib():
    global _declared
    L = oracle.lib()
    if not _declared:
        L.orc_ff_energy.argtypes = [ctypes.POINTER(_Batch), ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p]
        L.orc_ff_energy.restype = None
        L.orc_ff_gradient.argtypes = L.orc_ff_energy.argtypes
        L.orc_ff_gradient.restype = None
        L.orc_bfgs_minimize.argtypes = [ctypes.POINTER(_Batch), ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                        ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p]
        L.orc_bfgs_minimize.restype = None
        L.orc_etkdg_embed.argtypes = [ctypes.POINTER(_Molset), ctypes.POINTER(_Params), ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p]
        L.orc_etkdg_embed.restype = ctypes.c_int64
        L.orc_etkdg_random_coords.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
        L.orc_etkdg_random_coords.restype = None
        _declared = True
    return L


def _fill(c_groups, layout, groups, keep):
    for g, ((n_idx, n_par), (starts, idx, par)) in enumerate(zip(layout, groups)):
        starts = np.ascontiguousarray(starts, dtype=np.int32)
        idx = np.ascontiguousarray(np.asarray(idx, dtype=np.int32).reshape(-1, n_idx))
        par = np.ascontiguousarray(np.asarray(par, dtype=np.float64).reshape(len(idx), n_par))
        keep += [starts, idx, par]
        c_groups[g].starts = starts.ctypes.data
        c_groups[g].idx = idx.ctypes.data if idx.size else None
        c_groups[g].par = par.ctypes.data if par.size else None


class Batch:
    """Host-side twin of nvmolkit_amd.forcefield.FlatForcefieldBatch: same (starts, idx, par) groups, optional
    ``system_mol`` (term tables stored once per molecule)."""

    def __init__(self, kind: int, atom_starts, groups, system_mol=None, group_mask: int = 0):
        self.kind = kind
        self.dim = off.DIM[kind]
        self._keep = []
        self.atom_starts = np.ascontiguousarray(atom_starts, dtype=np.int32)
        self.n_systems = len(self.atom_starts) - 1
        self.c = _Batch()
        self.c.kind = kind
        self.c.n_systems = self.n_systems
        self.c.atom_starts = self.atom_starts.ctypes.data
        self.c.group_mask = group_mask
        if system_mol is not None:
            self.system_mol = np.ascontiguousarray(system_mol, dtype=np.int32)
            self.c.system_mol = self.system_mol.ctypes.data
        _fill(self.c.groups, off.LAYOUT[kind], groups, self._keep)

    def _active(self, active):
        if active is None:
            return None, None
        a = np.ascontiguousarray(active, dtype=np.uint8)
        return a, a.ctypes.data

    def energy(self, pos, w0: float = 1.0, w1: float = 1.0, active=None) -> np.ndarray:
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1)
        out = np.zeros(self.n_systems)
        keep, ap = self._active(active)
        _lib().orc_ff_energy(ctypes.byref(self.c), w0, w1, pos.ctypes.data, ap, out.ctypes.data)
        return out

    def gradient(self, pos, w0: float = 1.0, w1: float = 1.0, active=None) -> np.ndarray:
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1)
        out = np.zeros_like(pos)
        keep, ap = self._active(active)
        _lib().orc_ff_gradient(ctypes.byref(self.c), w0, w1, pos.ctypes.data, ap, out.ctypes.data)
        return out

    def minimize(self, pos, max_iters: int = 200, grad_tol: float = 1e-4, scale_grads: bool = True, w0: float = 1.0,
                 w1: float = 1.0, active=None):
        """Returns (positions, energies, statuses (0 = converged), iterations); the input array is not modified."""
        x = np.array(pos, dtype=np.float64).reshape(-1).copy()
        e = np.zeros(self.n_systems)
        st = np.full(self.n_systems, -1, dtype=np.int16)
        it = np.zeros(self.n_systems, dtype=np.int32)
        keep, ap = self._active(active)
        _lib().orc_bfgs_minimize(ctypes.byref(self.c), w0, w1, int(max_iters), float(grad_tol), int(bool(scale_grads)),
                                 x.ctypes.data, ap, e.ctypes.data, st.ctypes.data, it.ctypes.data)
        return x, e, st, it


def batch_from_systems(kind: int, systems, **kw) -> Batch:
    """systems: list of (pos, groups[(idx, par)]) as produced by nvmolkit_amd.synthetic / tests.util."""
    from nvmolkit_amd.synthetic import build_ff_batch_arrays

    a_s, flat, groups = build_ff_batch_arrays(kind, systems)
    return Batch(kind, a_s, groups, **kw), flat


def random_coords(seed: int, attempt: int, n_atoms: int, box: float) -> np.ndarray:
    out = np.zeros((n_atoms, 4))
    _lib().orc_etkdg_random_coords(seed & 0xFFFFFFFFFFFFFFFF, attempt, n_atoms, float(box), out.ctypes.data)
    return out


def etkdg_embed(mols, confs_per_molecule: int = 1, max_iterations: int = -1, batch_size: int = 4096,
                use_exp_torsions: bool = True, use_basic_knowledge: bool = True, enforce_chirality: bool = True,
                box_size_mult: float = 2.0, force_tol: float = 1e-3, seed: int = 42):
    """CPU twin of nvmolkit_amd.embedMolecules.embed_flat on a list of FlatMolecule-like objects (attributes n_atoms, dg,
    etk, checks, num_impropers).  Returns (coords flat float64, conf_counts, slot_starts, stage_failures, bfgs_iterations)."""
    n_atoms = np.array([m.n_atoms for m in mols], dtype=np.int32)
    if max_iterations == -1:
        max_iterations = 10 * int(n_atoms.max()) if len(n_atoms) else 1
    keep: list = [n_atoms]
    ms = _Molset()
    ms.n_mols = len(mols)
    ms.n_atoms = n_atoms.ctypes.data

    def stack(layout, per_mol):
        out = []
        for g, (n_idx, n_par) in enumerate(layout):
            starts = np.zeros(len(per_mol) + 1, dtype=np.int32)
            ii, pp = [], []
            for i, groups in enumerate(per_mol):
                idx = np.asarray(groups[g][0], dtype=np.int32).reshape(-1, n_idx)
                starts[i + 1] = starts[i] + len(idx)
                ii.append(idx)
                pp.append(np.asarray(groups[g][1], dtype=np.float64).reshape(len(idx), n_par))
            out.append((starts, np.concatenate(ii) if ii else np.zeros((0, n_idx), np.int32),
                        np.concatenate(pp) if pp else np.zeros((0, n_par))))
        return out

    _fill(ms.dg, off.LAYOUT[off.DG], stack(off.LAYOUT[off.DG], [m.dg for m in mols]), keep)
    has_etk = bool(mols) and all(m.etk is not None for m in mols)
    if has_etk:
        _fill(ms.etk, off.LAYOUT[off.ETK], stack(off.LAYOUT[off.ETK], [m.etk for m in mols]), keep)
    starts = np.zeros(len(mols) + 1, dtype=np.int32)
    kinds, idxs, pars = [], [], []
    for i, m in enumerate(mols):
        starts[i + 1] = starts[i] + len(m.checks)
        for kind, idx, par in m.checks:
            kinds.append(kind)
            idxs.append(list(idx) + [0] * (5 - len(idx)))
            pars.append(list(par) + [0.0] * (2 - len(par)))
    if kinds:
        ck = [starts, np.array(kinds, dtype=np.int32), np.array(idxs, dtype=np.int32), np.array(pars, dtype=np.float64)]
        keep += ck
        ms.check_starts, ms.check_kind, ms.check_idx, ms.check_par = (a.ctypes.data for a in ck)
    nimp = np.array([m.num_impropers for m in mols], dtype=np.int32)
    keep.append(nimp)
    ms.num_impropers = nimp.ctypes.data
    prm = _Params()
    prm.confs_per_mol = confs_per_molecule
    prm.max_iterations = max_iterations
    prm.batch_size = batch_size
    prm.use_exp_torsions = int(use_exp_torsions and has_etk)
    prm.use_basic_knowledge = int(use_basic_knowledge and has_etk)
    prm.enforce_chirality = int(enforce_chirality)
    prm.box_size = 5.0 * box_size_mult if box_size_mult > 0 else -box_size_mult
    prm.force_tol = force_tol
    prm.seed = seed & 0xFFFFFFFFFFFFFFFF
    slot = np.zeros(len(mols) + 1, dtype=np.int64)
    slot[1:] = np.cumsum(n_atoms.astype(np.int64) * confs_per_molecule * 3)
    coords = np.zeros(int(slot[-1]))
    counts = np.zeros(len(mols), dtype=np.int32)
    fails = np.zeros(N_STAGES, dtype=np.int32)
    iters = _lib().orc_etkdg_embed(ctypes.byref(ms), ctypes.byref(prm), coords.ctypes.data, counts.ctypes.data, fails.ctypes.data)
    return coords, counts, slot[:-1], fails, int(iters)
