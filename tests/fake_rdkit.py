"""A minimal stand-in for the slice of RDKit's Python API that the ingestion adapters call (there is no RDKit in the build
or GPU images).  ``install()`` registers fake ``rdkit`` modules; ``FakeMol.from_druglike`` wraps a molecule of
nvmolkit_amd.synthetic.druglike_molecule (graph + geometry + tables) so that the fake ``rdDistGeom`` functions can answer
with that molecule's own bounds matrix and torsion list.  Only what nvmolkit_amd/_rdkit_embed.py and the conformer drivers
use is provided."""

from __future__ import annotations

import contextlib
import sys
import types

import numpy as np


class _Enum:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return self.name


class _IntEnum(_Enum):
    def __init__(self, name, value):
        super().__init__(name)
        self.value = value

    def __int__(self):
        return self.value


BOND_TYPE_VALUES = {"SINGLE": 1, "DOUBLE": 2, "TRIPLE": 3, "AROMATIC": 12}  # RDKit::Bond::BondType


class FakeAtom:
    def __init__(self, mol, idx, z, tag="CHI_UNSPECIFIED", hyb="SP3", n_h=0, charge=0):
        self.mol, self.idx, self.z, self.tag, self.hyb, self.n_h, self.charge = mol, idx, z, tag, hyb, n_h, charge

    def GetNumExplicitHs(self):
        return 0

    def GetNumImplicitHs(self):
        return self.n_h

    def GetFormalCharge(self):
        return self.charge

    def GetMass(self):
        return 2.0 * self.z

    def GetIdx(self):
        return self.idx

    def GetAtomicNum(self):
        return self.z

    def GetChiralTag(self):
        return _Enum(self.tag)

    def GetDegree(self):
        return len(self.mol.nbrs[self.idx])

    def GetHybridization(self):
        return _Enum(self.hyb)


class FakeBond:
    def __init__(self, i, j, btype="SINGLE", stereo="STEREONONE", stereo_atoms=()):
        self.i, self.j, self.btype, self.stereo, self.stereo_atoms = i, j, btype, stereo, stereo_atoms

    def GetBeginAtomIdx(self):
        return self.i

    def GetEndAtomIdx(self):
        return self.j

    def GetBondType(self):
        if isinstance(self.btype, int):
            name = {v: k for k, v in BOND_TYPE_VALUES.items()}.get(self.btype, f"BT{self.btype}")
            return _IntEnum(name, self.btype)
        return _IntEnum(self.btype, BOND_TYPE_VALUES.get(self.btype, 0))

    def GetStereo(self):
        return _Enum(self.stereo)

    def GetStereoAtoms(self):
        return list(self.stereo_atoms)


class FakeRingInfo:
    def __init__(self, rings):
        self.rings = rings

    def AtomRingSizes(self, a):
        return [len(r) for r in self.rings if a in r]

    def NumAtomRings(self, a):
        return len(self.AtomRingSizes(a))

    def IsAtomInRingOfSize(self, a, k):
        return k in self.AtomRingSizes(a)


class FakeConformer:
    def __init__(self, n):
        self.xyz = np.zeros((n, 3))
        self.cid = -1

    def SetAtomPosition(self, a, p):
        self.xyz[a] = (p.x, p.y, p.z)

    def GetPositions(self):
        return self.xyz

    def SetPositions(self, xyz):
        self.xyz = np.array(xyz, dtype=np.float64)

    def GetId(self):
        return self.cid


class FakeMol:
    def __init__(self, atoms_z, bonds, rings=(), bounds=None, torsions=()):
        self.nbrs = [[] for _ in atoms_z]
        self.atoms = [FakeAtom(self, i, z) for i, z in enumerate(atoms_z)]
        self.bonds = []
        for b in bonds:
            b = b if isinstance(b, FakeBond) else FakeBond(*b)
            self.bonds.append(b)
            self.nbrs[b.i].append(b.j)
            self.nbrs[b.j].append(b.i)
        self.ring_info = FakeRingInfo([set(r) for r in rings])
        self.bounds, self.torsions = bounds, list(torsions)
        self.confs = []

    @classmethod
    def from_druglike(cls, m):
        """Wrap a synthetic.druglike_molecule() result: heavy atoms become carbons, cap-3 planar centres are SP2."""
        n = m["embed"]["n_atoms"]
        z = [6 if h else 1 for h in m["heavy"]]
        pairs, lb, ub = m["bounds"]
        bmat = np.zeros((n, n))
        bmat[pairs[:, 0], pairs[:, 1]] = ub          # upper triangle: upper bounds
        bmat[pairs[:, 1], pairs[:, 0]] = lb          # lower triangle: lower bounds
        tors = [dict(atomIndices=tuple(int(x) for x in idx), V=list(par[:6]), signs=[int(s) for s in par[6:]])
                for idx, par in zip(*m["embed"]["etk"][0])]
        mol = cls(z, [(int(i), int(j)) for i, j in m["bonds"]], bounds=bmat, torsions=tors)
        planar = {int(t[1]) for t in m["embed"]["etk"][1][0]}
        for a in planar:
            mol.atoms[a].hyb = "SP2"
        return mol

    def GetNumAtoms(self):
        return len(self.atoms)

    def GetNumBonds(self):
        return len(self.bonds)

    def GetAtoms(self):
        return self.atoms

    def GetAtomWithIdx(self, i):
        return self.atoms[i]

    def GetBonds(self):
        return self.bonds

    def GetRingInfo(self):
        return self.ring_info

    def RemoveAllConformers(self):
        self.confs = []

    def AddConformer(self, conf, assignId=True):
        conf.cid = len(self.confs)
        self.confs.append(conf)
        return conf.cid

    def GetNumConformers(self):
        return len(self.confs)

    def GetConformers(self):
        return list(self.confs)

    def GetConformer(self, cid=0):
        return self.confs[cid]


class FakeEmbedParameters:
    """The attributes of rdDistGeom.EmbedParameters that EmbedMolecules reads (ETKDGv3-like defaults)."""

    def __init__(self, **kw):
        self.useRandomCoords = True
        self.useExpTorsionAnglePrefs = True
        self.useBasicKnowledge = True
        self.enforceChirality = True
        self.ETversion = 2
        self.randomSeed = 42
        self.pruneRmsThresh = -1.0
        self.boxSizeMult = 2.0
        self.optimizerForceTol = 1e-3
        self.boundsMatForceScaling = 1.0
        self.ignoreSmoothingFailures = False
        self.useMacrocycle14config = True
        self.forceTransAmides = True
        self.onlyHeavyAtomsForRMS = False
        self.__dict__.update(kw)


@contextlib.contextmanager
def install():
    """Register fake ``rdkit`` modules for the duration of the block."""
    class Point3D:
        def __init__(self, x, y, z):
            self.x, self.y, self.z = x, y, z

    rdkit = types.ModuleType("rdkit")
    chem = types.ModuleType("rdkit.Chem")
    dg = types.ModuleType("rdkit.Chem.rdDistGeom")
    geom = types.ModuleType("rdkit.Geometry")

    def bounds(mol, set15bounds=True, scaleVDW=False, doTriangleSmoothing=True, useMacrocycle14config=False, **kw):
        if mol.bounds is None:
            raise RuntimeError("bounds smoothing failed")
        return mol.bounds.copy()

    dg.GetMoleculeBoundsMatrix = bounds
    dg.GetExperimentalTorsions = lambda mol, params=None: tuple(mol.torsions)
    class _Table:
        @staticmethod
        def GetAtomicWeight(z):
            return 2.0 * z

    chem.GetPeriodicTable = lambda: _Table()
    chem.Conformer = FakeConformer
    chem.rdDistGeom = dg
    geom.Point3D = Point3D
    rdkit.Chem, rdkit.Geometry = chem, geom
    saved = {k: sys.modules.get(k) for k in ("rdkit", "rdkit.Chem", "rdkit.Chem.rdDistGeom", "rdkit.Geometry")}
    sys.modules.update({"rdkit": rdkit, "rdkit.Chem": chem, "rdkit.Chem.rdDistGeom": dg, "rdkit.Geometry": geom})
    try:
        yield
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
