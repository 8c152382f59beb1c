"""Batched MMFF94 conformer optimisation on the GPU (reference API: nvmolkit/mmffOptimization.py:60-201).

``MMFFOptimizeMoleculesConfs`` keeps the reference's signature and error behaviour; the RDKit -> flattened-term
adapter below is the Python counterpart of ``constructForcefieldContribs``
(rdkit_extensions/mmff_flattened_builder.cpp:41-541).  It needs RDKit (atom typing and parameter tables are
RDKit's) and could not be exercised in the RDKit-less build / GPU images — the tested seam is
:func:`optimize_flat`.
"""

from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from nvmolkit_amd.forcefield import MMFF, FlatForcefieldBatch, MoleculeTermTables, PendingTermTables, minimize_device_conformers
from nvmolkit_amd.types import CoordinateOutput, Device3DResult, HardwareOptions

_LINEAR_MMFF_TYPES = frozenset({4, 53, 61})  # MMFFPROP.PAR rows with linh = 1 (CSP, =N=, NR%)
_TORSION_BOND_SMARTS = "[!$([D1]);!$([#1])]~[!$([D1]);!$([#1])]"  # RDKit DefaultTorsionBondSmarts


def optimize_flat(atom_starts, groups, positions: torch.Tensor, max_iters: int = 200, grad_tol: float = 1e-4,
                  system_mol=None):
    """Minimise flattened MMFF systems in place; returns (energies, converged) tensors.

    ``groups`` are the 7 MMFF term groups (include/nvmolkit_amd.h); with ``system_mol`` their rows are molecules and
    conformer s uses row ``system_mol[s]``.  gradTol 1e-4 is fixed on the reference's path
    (src/minimizer/bfgs_mmff.cpp:327)."""
    batch = FlatForcefieldBatch(MMFF, atom_starts, groups, device=positions.device, system_mol=system_mol)
    energies, statuses, _ = batch.minimize(positions, max_iters=max_iters, grad_tol=grad_tol, scale_grads=True)
    return energies, statuses == 0


def resident_tables(tables, device="cuda", preprocessing_threads: int = -1, wait: bool = True, after=None):
    """Assemble and upload the per-molecule MMFF term tables once (see :class:`MoleculeTermTables`); pass the result to
    :func:`optimize_device` instead of ``tables`` when the same molecules are optimised more than once.  ``wait=False`` returns at
    once with a :class:`PendingTermTables`: the tables are put together on a host thread and a side stream while the caller runs
    something else (the ETKDG embedding of the same molecules), and :func:`optimize_device` picks them up when it needs them."""
    if not wait:
        return PendingTermTables(MMFF, tables, device, preprocessing_threads, after=after)  # (after: see PendingTermTables)
    return MoleculeTermTables(MMFF, tables, device, preprocessing_threads)


def optimize_device(tables, conformers: Device3DResult, max_iters: int = 200, grad_tol: float = 1e-4) -> Device3DResult:
    """MMFF-minimise the conformers of a :class:`Device3DResult` (e.g. ``embed_flat(..., output=DEVICE)``) on the GPU
    they live on; ``tables[m]`` are the 7 MMFF term groups ``(idx, par)`` of input molecule m, or the
    :func:`resident_tables` of those molecules.  DEVICE in, DEVICE out
    (reference: MMFFOptimizeMoleculesConfs(..., output=DEVICE) fed by a ``deviceInput``)."""
    return minimize_device_conformers(MMFF, tables, conformers, max_iters, grad_tol)


def mmff_dielectric(props, dielectric_model=None, dielectric_constant=None):
    """(model code, constant) of the electrostatic term: 1 = constant dielectric, 2 = distance dependent (the codes
    ``mmff_ele`` takes; reference: addEle, rdkit_extensions/mmff_flattened_builder.cpp, divides the charge product by
    getMMFFDielectricConstant() and passes getMMFFDielectricModel()).  RDKit's Python ``MMFFMolProperties`` has SETTERS for
    these two but no getters, so non-default settings made with ``SetMMFFDielectricModel`` / ``SetMMFFDielectricConstant``
    cannot be read back: pass them explicitly (``dielectricModel`` / ``dielectricConstant`` of
    ``MMFFOptimizeMoleculesConfs``).  Getters are used when an RDKit build provides them."""
    model, const = dielectric_model, dielectric_constant
    if model is None:
        get = getattr(props, "GetMMFFDielectricModel", None)
        model = get() if callable(get) else 1
    if const is None:
        get = getattr(props, "GetMMFFDielectricConstant", None)
        const = get() if callable(get) else 1.0
    if isinstance(model, str):
        model = {"constant": 1, "distance": 2}[model.lower()]
    if isinstance(model, bool):  # RDKit's own flag: SetMMFFDielectricModel(distDepDielectric)
        model = 2 if model else 1
    if model not in (1, 2):
        raise ValueError("dielectricModel must be 'constant' / 1 or 'distance' / 2")
    const = float(const)
    if not const > 0.0:
        raise ValueError("the dielectric constant must be positive")
    return int(model), const


def flatten_mmff_from_rdkit(mol, props, conf_id: int = -1, non_bonded_threshold: float = 100.0,
                            ignore_interfrag_interactions: bool = True, dielectric_model=None, dielectric_constant=None):
    """RDKit molecule -> the 7 MMFF term groups (local atom indices).  UNTESTED without RDKit (see module docstring)."""
    from rdkit import Chem

    diel_model, diel_const = mmff_dielectric(props, dielectric_model, dielectric_constant)

    n = mol.GetNumAtoms()
    bonds, angles, strbend, oops, tors, vdw, ele = ([] for _ in range(7))
    for b in mol.GetBonds():  # addBonds, mmff_flattened_builder.cpp:41-59
        i, j = b.GetBeginAtomIdx(), b.GetEndAtomIdx()
        p = props.GetMMFFBondStretchParams(mol, i, j)
        if p:
            bonds.append((i, j, p[2], p[1]))  # r0, kb
    for j in range(n):  # addAngles :122-168, addStretchBend :170-237
        aj = mol.GetAtomWithIdx(j)
        if aj.GetDegree() == 1:
            continue
        linear = props.GetMMFFAtomType(j) in _LINEAR_MMFF_TYPES
        nbrs = [a.GetIdx() for a in aj.GetNeighbors()]
        for x in range(len(nbrs)):
            for y in range(x + 1, len(nbrs)):
                i, k = nbrs[x], nbrs[y]
                pa = props.GetMMFFAngleBendParams(mol, i, j, k)
                if pa:
                    angles.append((i, j, k, pa[2], pa[1], 1.0 if linear else 0.0))
                if linear:
                    continue
                ps = props.GetMMFFStretchBendParams(mol, i, j, k)
                b1, b2 = props.GetMMFFBondStretchParams(mol, i, j), props.GetMMFFBondStretchParams(mol, k, j)
                if ps and pa and b1 and b2:
                    strbend.append((i, j, k, pa[2], b1[2], b2[2], ps[1], ps[2]))
    for j in range(n):  # addOop :239-296: three permutations per trigonal centre
        aj = mol.GetAtomWithIdx(j)
        if aj.GetDegree() != 3:
            continue
        a, c, d = (x.GetIdx() for x in aj.GetNeighbors())
        koop = props.GetMMFFOopBendParams(mol, a, j, c, d)
        if koop is None:
            continue
        for i1, i3, i4 in ((a, c, d), (a, d, c), (c, d, a)):
            oops.append((i1, j, i3, i4, koop))
    query = Chem.MolFromSmarts(_TORSION_BOND_SMARTS)  # addTorsions :298-371
    sp23 = (Chem.HybridizationType.SP2, Chem.HybridizationType.SP3)
    for j, k in mol.GetSubstructMatches(query):
        aj, ak = mol.GetAtomWithIdx(j), mol.GetAtomWithIdx(k)
        if aj.GetHybridization() not in sp23 or ak.GetHybridization() not in sp23:
            continue
        for bi in aj.GetBonds():
            i = bi.GetOtherAtomIdx(j)
            if i == k:
                continue
            for bl in ak.GetBonds():
                l_ = bl.GetOtherAtomIdx(k)
                if l_ == j or l_ == i:
                    continue
                p = props.GetMMFFTorsionParams(mol, i, j, k, l_)
                if p:
                    tors.append((i, j, k, l_, p[1], p[2], p[3]))
    # non-bonded pairs: relation >= 1-4, within the threshold, same fragment (:373-470)
    dm = Chem.GetDistanceMatrix(mol)
    xyz = mol.GetConformer(conf_id).GetPositions()
    frags = np.zeros(n, dtype=int)
    if ignore_interfrag_interactions:
        for f, atoms in enumerate(Chem.GetMolFrags(mol)):
            frags[list(atoms)] = f
    charges = [props.GetMMFFPartialCharge(i) for i in range(n)]
    for i in range(n):
        for j in range(i + 1, n):
            if frags[i] != frags[j] or dm[i, j] < 3:
                continue
            if np.linalg.norm(xyz[i] - xyz[j]) > non_bonded_threshold:
                continue
            pv = props.GetMMFFVdWParams(i, j)
            if pv:
                vdw.append((i, j, pv[2], pv[3]))
            if abs(charges[i]) > 1e-10 and abs(charges[j]) > 1e-10:
                ele.append((i, j, charges[i] * charges[j] / diel_const, float(diel_model), 1.0 if dm[i, j] == 3 else 0.0))

    def split(rows, n_idx, n_par):
        a = np.array(rows, dtype=np.float64).reshape(-1, n_idx + n_par)
        return a[:, :n_idx].astype(np.int32), a[:, n_idx:]

    return [split(bonds, 2, 2), split(angles, 3, 3), split(strbend, 3, 5), split(oops, 4, 1), split(tors, 4, 3),
            split(vdw, 2, 2), split(ele, 2, 3)]


def MMFFOptimizeMoleculesConfs(molecules, maxIters: int = 200, properties=None, nonBondedThreshold=100.0,
                               ignoreInterfragInteractions=True, hardwareOptions: HardwareOptions | None = None,
                               output: CoordinateOutput = CoordinateOutput.RDKIT_CONFORMERS, targetGpu: int = -1,
                               dielectricModel=None, dielectricConstant=None):
    """Optimise every conformer of every molecule with MMFF94 + BFGS on the GPU.

    Same contract as the reference (nvmolkit/mmffOptimization.py:60-201): ``RDKIT_CONFORMERS`` updates the conformers
    in place and returns a list of per-conformer energies per molecule, ``DEVICE`` returns a :class:`Device3DResult`;
    ``ValueError(message, {"none": [...], "no_params": [...]})`` for ``None`` entries or molecules without MMFF
    parameters.  ``dielectricModel`` ("constant" / "distance") and ``dielectricConstant`` carry the two
    ``MMFFMolProperties`` settings that RDKit's Python API can set but not read back (see :func:`mmff_dielectric`);
    default: constant dielectric, D = 1, RDKit's defaults."""
    if not molecules:
        if output == CoordinateOutput.DEVICE:
            raise ValueError("MMFFOptimizeMoleculesConfs(output=DEVICE) requires at least one molecule")
        return []
    try:
        from rdkit.Chem import rdForceFieldHelpers as ffh
    except ImportError as exc:
        raise ImportError("MMFFOptimizeMoleculesConfs needs RDKit for MMFF typing; use optimize_flat() with "
                          "flattened term tables") from exc
    none_idx = [i for i, m in enumerate(molecules) if m is None]
    no_params = [i for i, m in enumerate(molecules) if m is not None and not ffh.MMFFHasAllMoleculeParams(m)]
    if none_idx or no_params:
        parts = []
        if none_idx:
            parts.append(f"None at indices {none_idx}")
        if no_params:
            parts.append(f"lacking MMFF atom types at indices {no_params}")
        raise ValueError("; ".join(parts), {"none": none_idx, "no_params": no_params})

    def per_mol(value, name):
        if isinstance(value, Sequence) and not isinstance(value, (str, bytes)) and not hasattr(value, "SetMMFFVariant"):
            if len(value) != len(molecules):
                raise ValueError(f"Expected {len(molecules)} values for {name}, got {len(value)}")
            return list(value)
        return [value] * len(molecules)

    props = [p if p is not None else ffh.MMFFGetMoleculeProperties(m) for m, p in zip(molecules, per_mol(properties, "properties"))]
    thresholds = per_mol(nonBondedThreshold, "nonBondedThreshold")
    interfrag = per_mol(ignoreInterfragInteractions, "ignoreInterfragInteractions")
    from nvmolkit_amd._rdkit_confs import optimize_rdkit_conformers

    return optimize_rdkit_conformers(
        MMFF, molecules,
        lambda mi, cid: flatten_mmff_from_rdkit(molecules[mi], props[mi], cid, float(thresholds[mi]), bool(interfrag[mi]),
                                                dielectricModel, dielectricConstant),
        int(maxIters), 1e-4, hardwareOptions, output, targetGpu)
