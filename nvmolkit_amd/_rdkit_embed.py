"""RDKit molecule -> flattened ETKDG tables (the ingestion side of ``EmbedMolecules``).

Host-side counterpart of the reference's ``prepareEmbedderArgs`` (src/embedder_utils.cpp:662-712: bounds matrix with
triangle smoothing, experimental torsions, chiral sets, double-bond lists) and of its term builders
(rdkit_extensions/dist_geom_flattened_builder.cpp:56-540), written against RDKit's PUBLIC Python API only:

  * bounds matrix               ``rdDistGeom.GetMoleculeBoundsMatrix`` (setTopolBounds + triangleSmoothBounds; the
                                relaxed retry of setupInitialBoundsMatrix — no 1-5 bounds, scaled vdW — on failure)
  * experimental torsions       ``rdDistGeom.GetExperimentalTorsions`` (atom indices, V, signs)
  * chiral / tetrahedral sets   atom chiral tags, degrees and ring info, as findChiralSets (:102-210)
  * double-bond lists           bond types and stereo, as findDoubleBonds (:617-663)
  * improper (planarity) terms  RDKit keeps these in CrystalFFDetails.improperAtoms, which has no Python accessor: they
                                are re-derived with the rule the ETKDG paper gives (sp2 C / N / O with three neighbours),
                                coefficients from calcInversionCoefficientsAndForceConstant (builder :178-230)

Everything here works on duck-typed objects that expose the handful of RDKit methods used (tests drive it with such
objects: there is no RDKit in the build or GPU images), so the logic is exercised even where RDKit is absent.
"""

from __future__ import annotations

import math
from typing import Sequence

import numpy as np

from nvmolkit_amd import _native

KNOWN_DIST_FORCE_CONSTANT = 100.0   # dist_geom_flattened_builder.cpp:17-20
KNOWN_DIST_TOL = 0.01
TRIPLE_BOND_MIN_ANGLE, TRIPLE_BOND_MAX_ANGLE = 179.0, 180.0
IMPROPER_TORSION_FORCE_SCALING = 10.0
_GROUP15_ANGLE = {15: 84.4339, 33: 86.9735, 51: 87.7047, 83: 90.0}


def inversion_coefficients(atomic_num: int, is_c_bound_to_o: bool):
    """(force constant, C0, C1, C2) of an improper term (builder :178-230; RDKit's UFF inversion parameters)."""
    if atomic_num in (6, 7, 8):
        c0, c1, c2 = 1.0, -1.0, 0.0
        k = 50.0 if is_c_bound_to_o else 6.0
    else:
        w = math.pi / 180.0 * _GROUP15_ANGLE.get(atomic_num, 1.0)
        c2 = 1.0
        c1 = -4.0 * math.cos(w)
        c0 = -(c1 * math.cos(w) + c2 * math.cos(2.0 * w))
        k = 22.0 / (c0 + c1 + c2)
    return k / 3.0, c0, c1, c2


def _name(x) -> str:
    """Enum-like value -> its name ('DOUBLE', 'CHI_TETRAHEDRAL_CW', ...), for RDKit enums and plain strings alike."""
    return getattr(x, "name", None) or str(x).rsplit(".", 1)[-1]


def bounds_matrix(mol, params) -> np.ndarray:
    """Smoothed bounds matrix: upper bounds above the diagonal, lower bounds below (RDKit convention).  Mirrors
    setupInitialBoundsMatrix (src/embedder_utils.cpp:272-330): normal topological bounds first; if smoothing fails, again
    without 1-5 bounds and with scaled van der Waals radii; if that fails too, raise unless ignoreSmoothingFailures."""
    from rdkit.Chem import rdDistGeom

    macro = bool(getattr(params, "useMacrocycle14config", False))
    trans = bool(getattr(params, "forceTransAmides", True))

    def get(set15, scale, smooth):
        try:
            return rdDistGeom.GetMoleculeBoundsMatrix(mol, set15bounds=set15, scaleVDW=scale, doTriangleSmoothing=smooth,
                                                      useMacrocycle14config=macro, forceTransAmides=trans)
        except TypeError:  # older RDKit: no forceTransAmides keyword
            return rdDistGeom.GetMoleculeBoundsMatrix(mol, set15bounds=set15, scaleVDW=scale, doTriangleSmoothing=smooth,
                                                      useMacrocycle14config=macro)

    try:
        return np.asarray(get(True, False, True), dtype=np.float64)
    except Exception:  # noqa: BLE001  (RDKit raises a generic exception when smoothing fails)
        try:
            return np.asarray(get(False, True, True), dtype=np.float64)
        except Exception:  # noqa: BLE001
            if getattr(params, "ignoreSmoothingFailures", False):
                return np.asarray(get(False, True, False), dtype=np.float64)
            raise ValueError("Could not triangle bounds smooth molecule.") from None


def experimental_torsions(mol, params):
    """[(i, j, k, l), V[6], signs[6]] from RDKit's torsion-preference tables (ForceFields::CrystalFF)."""
    from rdkit.Chem import rdDistGeom

    out = []
    for t in rdDistGeom.GetExperimentalTorsions(mol, params):
        v = list(t["V"]) + [0.0] * 6
        s = list(t["signs"]) + [0] * 6
        out.append((tuple(int(a) for a in t["atomIndices"]), v[:6], [float(x) for x in s[:6]]))
    return out


def topology(mol):
    """Neighbour lists, bonds (i, j, type name), angles (i, j, k, is_triple) in RDKit's setTopolBounds order."""
    n = mol.GetNumAtoms()
    nbrs = [[] for _ in range(n)]
    bonds = []
    btype = {}
    for b in mol.GetBonds():
        i, j = int(b.GetBeginAtomIdx()), int(b.GetEndAtomIdx())
        nbrs[i].append(j)
        nbrs[j].append(i)
        bonds.append((i, j))
        btype[(i, j)] = btype[(j, i)] = _name(b.GetBondType())
    angles = []
    for j in range(n):
        for x, i in enumerate(nbrs[j]):
            for k in nbrs[j][x + 1:]:
                triple = btype[(i, j)] == "TRIPLE" or btype[(j, k)] == "TRIPLE"
                angles.append((i, j, k, int(triple)))
    return nbrs, bonds, btype, angles


def chiral_sets(mol, nbrs):
    """(chiral centres, tetrahedral centres) as findChiralSets (src/embedder_utils.cpp:102-210): each entry is
    (centre, n1, n2, n3, n4, vol_lower, vol_upper, in_fused_small_rings)."""
    ring = mol.GetRingInfo()
    chiral, tetra = [], []
    for atom in mol.GetAtoms():
        z = atom.GetAtomicNum()
        if z == 1:
            continue
        tag = _name(atom.GetChiralTag())
        a = int(atom.GetIdx())
        specified = tag in ("CHI_TETRAHEDRAL_CW", "CHI_TETRAHEDRAL_CCW")
        if not (specified or (z in (6, 7) and atom.GetDegree() == 4)):
            continue
        nb = list(nbrs[a])
        if len(nb) < 3:
            raise ValueError("Cannot be a chiral center")
        lo, hi = 5.0, 100.0
        if len(nb) < 4:
            lo = 2.0  # three neighbours give lower volumes (RDKit github #5883)
            nb.append(a)
        small = sum(1 for sz in ring.AtomRingSizes(a) if sz < 5)
        if tag == "CHI_TETRAHEDRAL_CCW":
            chiral.append((a, *nb[:4], lo, hi, 0))
        elif tag == "CHI_TETRAHEDRAL_CW":
            chiral.append((a, *nb[:4], -hi, -lo, 0))
        elif not (ring.NumAtomRings(a) < 2 or ring.IsAtomInRingOfSize(a, 3)):
            tetra.append((a, *nb[:4], 0.0, 0.0, int(small > 1)))
    return chiral, tetra


def double_bonds(mol, nbrs, btype):
    """(doubleBondEnds [(nbr, atom, other)], stereoDoubleBonds [((a0, a1, a2, a3), sign)]) as findDoubleBonds (:617-663)."""
    ends, stereo = [], []
    deg = [len(x) for x in nbrs]
    for b in mol.GetBonds():
        if _name(b.GetBondType()) != "DOUBLE":
            continue
        i, j = int(b.GetBeginAtomIdx()), int(b.GetEndAtomIdx())
        for atm, oatm in ((i, j), (j, i)):
            if deg[atm] < 2:
                continue
            for nbr in nbrs[atm]:
                if nbr == oatm:
                    continue
                if btype[(atm, nbr)] != "SINGLE" and deg[atm] == 2:
                    continue
                ends.append((nbr, atm, oatm))
        st = _name(b.GetStereo())
        if st in ("STEREOZ", "STEREOE", "STEREOCIS", "STEREOTRANS"):
            sa = [int(x) for x in b.GetStereoAtoms()]
            stereo.append(((sa[0], i, j, sa[1]), -1 if st in ("STEREOCIS", "STEREOZ") else 1))
    return ends, stereo


def improper_atoms(mol, nbrs, btype):
    """[(n0, centre, n1, n2, atomic number, isCBoundToO)]: sp2 C / N / O centres with three neighbours (the planarity
    terms of ETKDG's basic knowledge; RDKit: CrystalFFDetails.improperAtoms, no Python accessor)."""
    out = []
    for atom in mol.GetAtoms():
        z, a = atom.GetAtomicNum(), int(atom.GetIdx())
        if z not in (6, 7, 8) or len(nbrs[a]) != 3 or _name(atom.GetHybridization()) != "SP2":
            continue
        # RDKit's rule (the UFF inversion's isBoundToSP2O): a neighbouring oxygen whose own hybridisation is SP2 — the carbonyl
        # O, but also the ring O of a furan or the O of an ester / phenol conjugated with the centre
        bound_to_o = z == 6 and any(mol.GetAtomWithIdx(n).GetAtomicNum() == 8 and _name(mol.GetAtomWithIdx(n).GetHybridization()) == "SP2"
                                    for n in nbrs[a])
        out.append((nbrs[a][0], a, nbrs[a][1], nbrs[a][2], z, bool(bound_to_o)))
    return out


def flatten_etkdg_tables(n: int, bmat: np.ndarray, bonds, angles, torsions, impropers, chiral, tetra, dbl_ends, dbl_stereo,
                         use_exp_torsions: bool = True, use_basic_knowledge: bool = True, bounds_force_scaling: float = 1.0):
    """Chemistry lists -> ``FlatMolecule`` fields (term layouts of include/nvmolkit_amd.h).  Pure numpy: this is the part
    the reference does in constructForceFieldContribs / construct3DForceFieldContribs (builder :472-540)."""
    ub = np.triu(bmat, 1)
    ub = ub + ub.T
    lb = np.tril(bmat, -1)
    lb = lb + lb.T
    iu = np.triu_indices(n, 1)
    pairs = np.stack([iu[1], iu[0]], 1).astype(np.int64)  # (i, j) with i > j like addDistViolationContribs
    dg = [(pairs, np.stack([lb[iu] ** 2, ub[iu] ** 2, np.ones(len(pairs))], 1)),
          (np.array([c[1:5] for c in chiral], dtype=np.int64).reshape(-1, 4),
           np.array([c[5:7] for c in chiral], dtype=np.float64).reshape(-1, 2)),
          (np.arange(n, dtype=np.int64).reshape(-1, 1), np.zeros((n, 0)))]
    checks = []
    for c in tetra:
        checks.append((_native.CHECK_TETRAHEDRAL, c[0:5], (float(c[7]),)))
    chiral_idx = set()
    for c in chiral:
        checks.append((_native.CHECK_CHIRAL_VOLUME, (0,) + tuple(c[1:5]), (c[5], c[6])))
        checks.append((_native.CHECK_CHIRAL_CENTER_VOLUME, c[0:5], ()))
        if c[0] != c[4]:
            chiral_idx.update(c[0:5])
    cl = sorted(chiral_idx)
    for x in range(len(cl)):
        for y in range(x + 1, len(cl)):
            checks.append((_native.CHECK_CHIRAL_DISTANCE, (cl[x], cl[y]), (lb[cl[x], cl[y]], ub[cl[x], cl[y]])))
    for e in dbl_ends:
        checks.append((_native.CHECK_DOUBLE_BOND_GEOMETRY, tuple(e), ()))
    for idx, sign in dbl_stereo:
        checks.append((_native.CHECK_DOUBLE_BOND_STEREO, tuple(idx), (float(sign),)))

    etk, n_imp = None, 0
    if use_exp_torsions or use_basic_knowledge:
        seen = np.zeros((n, n), dtype=bool)

        def mark(i, j):
            seen[min(i, j), max(i, j)] = True

        t_idx, t_par = [], []
        if use_exp_torsions:
            for (i, j, k, l), v, s in torsions:
                if len({i, j, k, l}) != 4:
                    raise ValueError("degenerate points")
                mark(i, l)
                t_idx.append((i, j, k, l))
                t_par.append(list(v) + list(s))
        imp_idx, imp_par = [], []
        constrained = np.zeros(n, dtype=bool)
        if use_basic_knowledge:
            n_imp = len(impropers)
            for (a0, a1, a2, a3, z, c_o) in impropers:
                k, c0, c1, c2 = inversion_coefficients(z, c_o)
                for (p, q, r) in ((a0, a2, a3), (a0, a3, a2), (a2, a3, a0)):  # the three permutations of builder :242-268
                    imp_idx.append((p, a1, q, r))
                    imp_par.append((c0, c1, c2, k * IMPROPER_TORSION_FORCE_SCALING))
                constrained[a1] = True
        d12_idx, d12_par = [], []
        for (i, j) in bonds:  # distance +- 0.01 around the CURRENT distance (re-centred on the device: par[3] = 0)
            mark(i, j)
            d12_idx.append((i, j))
            d12_par.append((-KNOWN_DIST_TOL, KNOWN_DIST_TOL, KNOWN_DIST_FORCE_CONSTANT, 0.0))
        d13_idx, d13_par, ang_idx, ang_par = [], [], [], []
        for (i, j, k, triple) in angles:
            mark(i, k)
            if use_basic_knowledge and triple:
                ang_idx.append((i, j, k))
                ang_par.append((TRIPLE_BOND_MIN_ANGLE, TRIPLE_BOND_MAX_ANGLE))
            elif constrained[j]:  # bounds-matrix window, pinned (isImproperConstrained)
                d13_idx.append((i, k))
                d13_par.append((lb[i, k], ub[i, k], KNOWN_DIST_FORCE_CONSTANT, 1.0))
            else:
                d13_idx.append((i, k))
                d13_par.append((-KNOWN_DIST_TOL, KNOWN_DIST_TOL, KNOWN_DIST_FORCE_CONSTANT, 0.0))
        rest = ~seen[iu]
        lr = np.stack([iu[1][rest], iu[0][rest]], 1).astype(np.int64)
        lr_par = np.stack([lb[iu][rest], ub[iu][rest], np.full(int(rest.sum()), 10.0 * bounds_force_scaling),
                           np.zeros(int(rest.sum()))], 1)
        a2 = lambda x, w: np.array(x, dtype=np.int64).reshape(-1, w)  # noqa: E731
        f2 = lambda x, w: np.array(x, dtype=np.float64).reshape(-1, w)  # noqa: E731
        etk = [(a2(t_idx, 4), f2(t_par, 12)), (a2(imp_idx, 4), f2(imp_par, 4)), (a2(d12_idx, 2), f2(d12_par, 4)),
               (a2(d13_idx, 2), f2(d13_par, 4)), (a2(ang_idx, 3), f2(ang_par, 2)), (lr, lr_par)]
    return dict(n_atoms=n, dg=dg, etk=etk, checks=checks, num_impropers=n_imp)


def flatten_etkdg_from_rdkit(mol, params) -> dict:
    """One RDKit molecule -> the fields of a :class:`nvmolkit_amd.embedMolecules.FlatMolecule`."""
    n = mol.GetNumAtoms()
    if n == 0:
        raise ValueError("molecule has no atoms")
    et_version = int(getattr(params, "ETversion", 1))
    if et_version < 1 or et_version > 2:
        raise ValueError("Only version 1 and 2 of the experimental torsion-angle preferences (ETversion) supported")
    use_et = bool(getattr(params, "useExpTorsionAnglePrefs", False))
    use_bk = bool(getattr(params, "useBasicKnowledge", False))
    nbrs, bonds, btype, angles = topology(mol)
    bmat = bounds_matrix(mol, params)
    torsions = experimental_torsions(mol, params) if use_et else []
    impropers = improper_atoms(mol, nbrs, btype) if use_bk else []
    chiral, tetra = chiral_sets(mol, nbrs)
    ends, stereo = double_bonds(mol, nbrs, btype)
    return flatten_etkdg_tables(n, bmat, bonds, angles, torsions, impropers, chiral, tetra, ends, stereo, use_et, use_bk,
                                float(getattr(params, "boundsMatForceScaling", 1.0)))


def write_conformers(mol, coords: np.ndarray, clear: bool = True) -> Sequence[int]:
    """(k, n_atoms, 3) coordinates -> conformers of `mol` (ids 0..k-1), like the reference's RDKIT_CONFORMERS output."""
    from rdkit import Chem
    from rdkit.Geometry import Point3D

    if clear:
        mol.RemoveAllConformers()
    ids = []
    for xyz in coords:
        conf = Chem.Conformer(mol.GetNumAtoms())
        for a, (x, y, z) in enumerate(xyz):
            conf.SetAtomPosition(a, Point3D(float(x), float(y), float(z)))
        ids.append(mol.AddConformer(conf, assignId=True))
    return ids


def self_matches_for_pruning(mol, symmetrize_terminal_groups: bool = True, max_matches: int = 1000) -> np.ndarray:
    """(K, n_heavy) atom indices of ``mol`` for symmetry-aware RMS pruning, row 0 the heavy atoms themselves and rows k > 0 their
    images under the molecule's other self matches — RDKit's own substructure search on the hydrogen-stripped molecule, as the
    reference asks for it (getMolSelfMatches, rdkit_extensions/conformer_pruning.cpp:24-60).  With
    ``symmetrize_terminal_groups`` the query copy has its conjugated terminal N / O groups made alike (what
    MolAlign::details::symmetrizeTerminalAtoms does in C++, which RDKit does not export to Python: the atoms become
    atomic-number queries and their bonds single-or-double queries)."""
    from rdkit import Chem

    ps = Chem.RemoveHsParameters()
    stripped = Chem.RemoveHs(mol, ps, sanitize=False)
    query = Chem.RWMol(stripped)
    if symmetrize_terminal_groups:
        pattern = Chem.MolFromSmarts("[O,N;D1;$([O,N;D1]-[*]=[O,N;D1]),$([O,N;D1]=[*]-[O,N;D1])]~[*]")
        either = Chem.MolFromSmarts("*-,=*").GetBondWithIdx(0)
        for end, centre in stripped.GetSubstructMatches(pattern, uniquify=False):
            query.ReplaceAtom(end, Chem.AtomFromSmarts(f"[#{stripped.GetAtomWithIdx(end).GetAtomicNum()}]"))
            query.ReplaceBond(query.GetBondBetweenAtoms(end, centre).GetIdx(), either)
    to_full = mol.GetSubstructMatches(query, uniquify=True, maxMatches=1)
    if len(to_full) != 1:
        raise RuntimeError("the hydrogen-stripped molecule was not found in the molecule")
    found = stripped.GetSubstructMatches(query, uniquify=False, maxMatches=int(max_matches))
    full = np.asarray(to_full[0], dtype=np.int32)
    return np.stack([full[np.asarray(m, dtype=np.int64)] for m in found]).astype(np.int32)
