"""Batched UFF conformer optimisation on the GPU (reference API: nvmolkit/uffOptimization.py:30-142).

``UFFOptimizeMoleculesConfs`` keeps the reference's signature and error behaviour.  The RDKit -> flattened-term
adapter is the Python counterpart of ``constructForcefieldContribs`` (rdkit_extensions/uff_flattened_builder.cpp:
139-560); it needs RDKit (atom typing and the UFF parameter table are RDKit's) and could not be
exercised in the RDKit-less build / GPU images — the tested seams are :func:`optimize_flat` and
:func:`optimize_device`.
"""

from __future__ import annotations

import math
from typing import Sequence

import numpy as np
import torch

from nvmolkit_amd.forcefield import UFF, FlatForcefieldBatch, MoleculeTermTables, PendingTermTables, minimize_device_conformers
from nvmolkit_amd.types import CoordinateOutput, Device3DResult, HardwareOptions

_TORSION_BOND_SMARTS = "[!$([D1]);!$([#1])]~[!$([D1]);!$([#1])]"  # RDKit DefaultTorsionBondSmarts
_GROUP6 = frozenset({8, 16, 34, 52, 84})
_INVERSION_W0_DEG = {15: 84.4339, 33: 86.9735, 51: 87.7047, 83: 90.0}


def optimize_flat(atom_starts, groups, positions: torch.Tensor, max_iters: int = 1000, grad_tol: float = 1e-4,
                  system_mol=None):
    """Minimise flattened UFF systems in place; returns (energies, converged) tensors.

    ``groups`` are the 5 UFF term groups (include/nvmolkit_amd.h: bond, angle, torsion, inversion, vdW); with
    ``system_mol`` their rows are molecules.  gradTol 1e-4 as in UFFOptimizeMoleculesConfsBfgs
    (src/minimizer/bfgs_uff.cpp:257-263)."""
    batch = FlatForcefieldBatch(UFF, atom_starts, groups, device=positions.device, system_mol=system_mol)
    energies, statuses, _ = batch.minimize(positions, max_iters=max_iters, grad_tol=grad_tol, scale_grads=True)
    return energies, statuses == 0


def resident_tables(tables, device="cuda", preprocessing_threads: int = -1, wait: bool = True):
    """Assemble and upload the per-molecule UFF term tables once (see :class:`MoleculeTermTables`); pass the result to
    :func:`optimize_device` instead of ``tables`` when the same molecules are optimised more than once.  ``wait=False`` returns at
    once with a :class:`PendingTermTables`: the tables are put together on a host thread and a side stream while the caller runs
    something else (the ETKDG embedding of the same molecules), and :func:`optimize_device` picks them up when it needs them."""
    if not wait:
        return PendingTermTables(UFF, tables, device, preprocessing_threads)
    return MoleculeTermTables(UFF, tables, device, preprocessing_threads)


def optimize_device(tables, conformers: Device3DResult, max_iters: int = 1000, grad_tol: float = 1e-4) -> Device3DResult:
    """UFF-minimise the conformers of a :class:`Device3DResult` on the GPU they live on (DEVICE in, DEVICE out)."""
    return minimize_device_conformers(UFF, tables, conformers, max_iters, grad_tol)


def _angle_coefficients(theta0: float):
    """C0, C1, C2 of the order-0 (general) bend, uff_flattened_builder.cpp:62-70."""
    s, c = math.sin(theta0), math.cos(theta0)
    c2 = 1.0 / (4.0 * max(s * s, 1.0e-8))
    return c2 * (2.0 * c * c + 1.0), -4.0 * c2 * c, c2


def _torsion_shape(bond_order, z2, z3, sp3_2, sp3_3, end_atom_is_sp2):
    """(order, cosTerm) of a torsion about bond 2-3 (calcTorsionParams, uff_flattened_builder.cpp:86-136); the force
    constant comes from RDKit's own getter."""
    if sp3_2 and sp3_3:
        if bond_order == 1.0 and z2 in _GROUP6 and z3 in _GROUP6:
            return 2, -1.0
        return 3, -1.0
    if not sp3_2 and not sp3_3:
        return 2, 1.0
    if bond_order == 1.0:
        if (sp3_2 and z2 in _GROUP6 and z3 not in _GROUP6) or (sp3_3 and z3 in _GROUP6 and z2 not in _GROUP6):
            return 2, -1.0
        if end_atom_is_sp2:
            return 3, -1.0
    return 6, 1.0


def flatten_uff_from_rdkit(mol, conf_id: int = -1, vdw_threshold: float = 10.0, ignore_interfrag_interactions: bool = True):
    """RDKit molecule -> the 5 UFF term groups (local atom indices).  UNTESTED without RDKit (see module docstring).

    Force constants and rest values come from RDKit's parameter getters (``rdForceFieldHelpers.GetUFF*Params``); term
    enumeration, angle orders, torsion periodicities and inversion coefficients follow the reference builder.
    Centres the getters cannot describe (sp2 atoms in 3- / 4-membered rings, whose rest angle is overridden, and
    5-coordinate sp3d centres) raise ``NotImplementedError`` rather than produce approximate terms."""
    from rdkit import Chem
    from rdkit.Chem import rdForceFieldHelpers as ffh

    n = mol.GetNumAtoms()
    ring = mol.GetRingInfo()
    hyb = Chem.HybridizationType
    bonds, angles, tors, invs, vdw = ([] for _ in range(5))
    for b in mol.GetBonds():  # addBonds :139-154
        i, j = b.GetBeginAtomIdx(), b.GetEndAtomIdx()
        p = ffh.GetUFFBondStretchParams(mol, i, j)
        if p:
            bonds.append((i, j, p[1], p[0]))  # restLen, k
    for j in range(n):  # addAngles :156-228
        aj = mol.GetAtomWithIdx(j)
        if aj.GetDegree() == 1:
            continue
        hj = aj.GetHybridization()
        if hj == hyb.SP3D and aj.GetDegree() == 5:
            raise NotImplementedError("UFF trigonal-bipyramidal centres need RDKit's internal atomic parameters")
        if hj == hyb.SP2 and (ring.IsAtomInRingOfSize(j, 3) or ring.IsAtomInRingOfSize(j, 4)):
            raise NotImplementedError("UFF sp2 centres in 3-/4-membered rings need RDKit's internal atomic parameters")
        order = {hyb.SP: 1, hyb.SP2: 3, hyb.SP3D2: 4}.get(hj, 0)
        nbrs = [a.GetIdx() for a in aj.GetNeighbors()]
        for x in range(len(nbrs)):
            for y in range(x + 1, len(nbrs)):
                p = ffh.GetUFFAngleBendParams(mol, nbrs[x], j, nbrs[y])
                if not p:
                    continue
                theta0 = math.radians(p[1])
                c0, c1, c2 = _angle_coefficients(theta0) if order == 0 else (0.0, 0.0, 0.0)
                angles.append((nbrs[x], j, nbrs[y], theta0, p[0], order, c0, c1, c2))
    query = Chem.MolFromSmarts(_TORSION_BOND_SMARTS)  # addTorsions :381-466
    sp23 = (hyb.SP2, hyb.SP3)
    for j, k in mol.GetSubstructMatches(query):
        aj, ak = mol.GetAtomWithIdx(j), mol.GetAtomWithIdx(k)
        if aj.GetHybridization() not in sp23 or ak.GetHybridization() not in sp23:
            continue
        bond = mol.GetBondBetweenAtoms(j, k)
        rows = []
        for bi in aj.GetBonds():
            i = bi.GetOtherAtomIdx(j)
            if i == k:
                continue
            for bl in ak.GetBonds():
                l_ = bl.GetOtherAtomIdx(k)
                if l_ == j or l_ == i:
                    continue
                v = ffh.GetUFFTorsionParams(mol, i, j, k, l_)
                if v is None:
                    continue
                end_sp2 = hyb.SP2 in (mol.GetAtomWithIdx(i).GetHybridization(), mol.GetAtomWithIdx(l_).GetHybridization())
                order, cos_term = _torsion_shape(bond.GetBondTypeAsDouble(), aj.GetAtomicNum(), ak.GetAtomicNum(),
                                                 aj.GetHybridization() == hyb.SP3, ak.GetHybridization() == hyb.SP3, end_sp2)
                rows.append((i, j, k, l_, float(v), order, cos_term))
        tors.extend((i, j, k, l_, v / len(rows), o, c) for i, j, k, l_, v, o, c in rows)  # V split over the bond's torsions
    for j in range(n):  # addInversions :468-535: three permutations per trigonal centre
        aj = mol.GetAtomWithIdx(j)
        z = aj.GetAtomicNum()
        if z not in (6, 7, 8, 15, 33, 51, 83) or aj.GetDegree() != 3:
            continue
        if z in (6, 7, 8) and aj.GetHybridization() != hyb.SP2:
            continue
        a, c, d = (x.GetIdx() for x in aj.GetNeighbors())
        if z in (6, 7, 8):
            c0, c1, c2 = 1.0, -1.0, 0.0
        else:
            w0 = math.radians(_INVERSION_W0_DEG[z])
            c2 = 1.0
            c1 = -4.0 * math.cos(w0)
            c0 = -(c1 * math.cos(w0) + c2 * math.cos(2.0 * w0))
        for i1, i3, i4 in ((a, c, d), (a, d, c), (c, d, a)):
            kinv = ffh.GetUFFInversionParams(mol, i1, j, i3, i4)
            if kinv is not None:
                invs.append((i1, j, i3, i4, float(kinv), c0, c1, c2))
    dm = Chem.GetDistanceMatrix(mol)  # addNonbonded :335-379: relation >= 1-4, inside vdwThresh * x_ij, same fragment
    xyz = mol.GetConformer(conf_id).GetPositions()
    frags = np.zeros(n, dtype=int)
    if ignore_interfrag_interactions:
        for f, atoms in enumerate(Chem.GetMolFrags(mol)):
            frags[list(atoms)] = f
    for i in range(n):
        for j in range(i + 1, n):
            if frags[i] != frags[j] or dm[i, j] < 3:
                continue
            p = ffh.GetUFFVdWParams(mol, i, j)
            if not p:
                continue
            threshold = vdw_threshold * p[0]
            if np.linalg.norm(xyz[i] - xyz[j]) < threshold:
                vdw.append((i, j, p[0], p[1], threshold))

    def split(rows, n_idx, n_par):
        a = np.array(rows, dtype=np.float64).reshape(-1, n_idx + n_par)
        return a[:, :n_idx].astype(np.int32), a[:, n_idx:]

    return [split(bonds, 2, 2), split(angles, 3, 6), split(tors, 4, 3), split(invs, 4, 4), split(vdw, 2, 3)]


def UFFOptimizeMoleculesConfs(molecules, maxIters: int = 1000, vdwThreshold=10.0, ignoreInterfragInteractions=True,
                              hardwareOptions: HardwareOptions | None = None,
                              output: CoordinateOutput = CoordinateOutput.RDKIT_CONFORMERS, targetGpu: int = -1):
    """Optimise every conformer of every molecule with UFF + BFGS on the GPU.

    Same contract as the reference (nvmolkit/uffOptimization.py:30-142): ``RDKIT_CONFORMERS`` updates the conformers
    in place and returns per-molecule lists of energies, ``DEVICE`` returns a :class:`Device3DResult`;
    ``ValueError(message, {"none": [...], "no_params": [...]})`` for ``None`` entries or molecules lacking UFF atom
    types."""
    if not molecules:
        if output == CoordinateOutput.DEVICE:
            raise ValueError("UFFOptimizeMoleculesConfs(output=DEVICE) requires at least one molecule")
        return []
    try:
        from rdkit.Chem import rdForceFieldHelpers as ffh
    except ImportError as exc:
        raise ImportError("UFFOptimizeMoleculesConfs needs RDKit for UFF typing; use optimize_flat() / "
                          "optimize_device() with flattened term tables") from exc
    none_idx = [i for i, m in enumerate(molecules) if m is None]
    no_params = [i for i, m in enumerate(molecules) if m is not None and not ffh.UFFHasAllMoleculeParams(m)]
    if none_idx or no_params:
        parts = []
        if none_idx:
            parts.append(f"None at indices {none_idx}")
        if no_params:
            parts.append(f"lacking UFF atom types at indices {no_params}")
        raise ValueError("; ".join(parts), {"none": none_idx, "no_params": no_params})

    def per_mol(value, name):
        if isinstance(value, Sequence) and not isinstance(value, (str, bytes)):
            if len(value) != len(molecules):
                raise ValueError(f"Expected {len(molecules)} values for {name}, got {len(value)}")
            return list(value)
        return [value] * len(molecules)

    thresholds = [float(v) for v in per_mol(vdwThreshold, "vdwThreshold")]
    interfrag = [bool(v) for v in per_mol(ignoreInterfragInteractions, "ignoreInterfragInteractions")]
    from nvmolkit_amd._rdkit_confs import optimize_rdkit_conformers

    return optimize_rdkit_conformers(
        UFF, molecules, lambda mi, cid: flatten_uff_from_rdkit(molecules[mi], cid, thresholds[mi], interfrag[mi]),
        int(maxIters), 1e-4, hardwareOptions, output, targetGpu)
