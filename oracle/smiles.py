"""TEST INFRASTRUCTURE — an independent restatement of the SMILES ingestion of the fingerprint path.

The product's ingestion is C++ (nvmolkit_amd/csrc/smiles.cpp: character-level parser, low-link bridge search).  This file
derives the same quantities a second way (regular-expression tokens, ring bonds by removing each bond and asking whether its
ends stay connected, valence rules written from RDKit's documented model) so that the two can only agree by being right
about the rules, not by sharing code.  What is restated from the reference: the invariant components and their order
(src/morgan_fingerprint_common.cpp:80-121: Z, degree + Hs, Hs incl. hydrogen neighbours, formal charge,
int(mass - average mass), + [1] for ring atoms) and the bond invariant (RDKit bond type, :100).  RDKit itself is in
neither image: parity with RDKit's SMILES parser is UNPINNED except for the element-count known answers of
tests/test_morgan_fingerprint_ref.cpp:44-69, which go through this module in tests/test_smiles_ingestion.py.

Only tests/ may import this module.
"""

from __future__ import annotations

import re

import numpy as np

ELEMENTS = ("* H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr Nb Mo "
            "Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt Au Hg Tl Pb "
            "Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt Ds Rg Cn Nh Fl Mc Lv Ts Og").split()
Z_OF = {s: z for z, s in enumerate(ELEMENTS)}
AROMATIC_SYMBOLS = {"b": 5, "c": 6, "n": 7, "o": 8, "p": 15, "s": 16, "se": 34, "as": 33, "te": 52, "si": 14}
VALENCES = {5: (3,), 6: (4,), 7: (3,), 8: (2,), 9: (1,), 15: (3, 5, 7), 16: (2, 4, 6), 17: (1,), 35: (1,), 53: (1, 3, 5)}
# "/" and "\\" carry a direction, not an order: between two aromatic ring atoms the bond is aromatic, else single (= unmarked)
BOND_TYPE = {"-": 1, "/": None, "\\": None, "=": 2, "#": 3, "$": 4, ":": 12}
# average weights and a few exact isotope masses: only what the hand-made test molecules use (H, C, N, O, F, I)
WEIGHT = {0: 0.0, 1: 1.008, 6: 12.011, 7: 14.007, 8: 15.999, 9: 18.998, 53: 126.90}
ISOTOPE = {(1, 2): 2.01410, (1, 3): 3.01605, (6, 13): 13.00335, (6, 14): 14.00324, (7, 15): 15.00011, (8, 18): 17.99916,
           (9, 18): 18.00094, (53, 125): 124.90463, (53, 131): 130.90612}

TOKEN = re.compile(r"""
    (?P<bracket>\[[^\]]*\]) | (?P<organic>Cl|Br|[BCNOFPSI]|[bcnops]|\*) | (?P<bond>[-=#$:/\\]) |
    (?P<ring>%\(\d{1,5}\)|%\d\d|\d) | (?P<open>\() | (?P<close>\)) | (?P<dot>\.)""", re.X)
BRACKET = re.compile(r"^\[(?P<iso>\d+)?(?P<sym>se|as|te|si|[bcnops]|[A-Z][a-z]?|\*|\#\d+)(?P<chiral>@(?:@|(?:TH|AL|SP|TB|OH)\d+)?)?"
                     r"(?P<h>H\d?)?(?P<charge>\++\d*|-+\d*)?(?::\d+)?\]$")


class SmilesError(ValueError):
    pass


def parse(smiles: str):
    """-> (atoms, bonds): atoms are dicts (z, charge, isotope, h_explicit, bracket, aromatic), bonds [a, b, type or None]."""
    atoms, bonds, _ = parse_with_directions(smiles)
    return atoms, bonds


def parse_with_directions(smiles: str):
    """parse() plus the set of bond indices written with '/' or '\\'."""
    atoms, bonds, directional = [], [], set()
    prev, pending, stack, open_rings = None, None, [], {}
    pos = 0
    while pos < len(smiles):
        m = TOKEN.match(smiles, pos)
        if m is None:
            raise SmilesError(f"unexpected character {smiles[pos]!r} at {pos}")
        pos = m.end()
        kind, text = m.lastgroup, m.group()
        if kind in ("bracket", "organic"):
            if kind == "bracket":
                b = BRACKET.match(text)
                if b is None:
                    raise SmilesError(f"bad bracket atom {text}")
                sym = b["sym"]
                by_number = sym.startswith("#")  # "[#6]": the element by atomic number
                if by_number and int(sym[1:]) > 118:
                    raise SmilesError(f"atomic number out of range in {text}")
                aromatic = sym in AROMATIC_SYMBOLS
                if not aromatic and not by_number and sym not in Z_OF:
                    raise SmilesError(f"unknown element {sym}")
                h = 0 if b["h"] is None else (int(b["h"][1:]) if len(b["h"]) > 1 else 1)
                q = 0
                if b["charge"]:
                    c = b["charge"]
                    digits = c.lstrip("+-")
                    q = int(digits) if digits else len(c)
                    if c[0] == "-":
                        q = -q
                if int(b["iso"] or 0) > 999 or abs(q) > 15:
                    raise SmilesError(f"isotope or charge out of range in {text}")
                atoms.append(dict(z=int(sym[1:]) if by_number else (AROMATIC_SYMBOLS[sym] if aromatic else Z_OF[sym]), charge=q, isotope=int(b["iso"] or 0),
                                  h_explicit=h, bracket=True, aromatic=aromatic))
            else:
                aromatic = text.islower()
                atoms.append(dict(z=AROMATIC_SYMBOLS[text] if aromatic else Z_OF[text], charge=0, isotope=0, h_explicit=0,
                                  bracket=False, aromatic=aromatic))
            if prev is not None:
                if pending in ("/", "\\"):
                    directional.add(len(bonds))
                bonds.append([prev, len(atoms) - 1, BOND_TYPE.get(pending)])
            prev, pending = len(atoms) - 1, None
        elif kind == "bond":
            if prev is None or pending is not None:
                raise SmilesError("misplaced bond symbol")
            pending = text
        elif kind == "ring":
            if prev is None:
                raise SmilesError("ring closure before any atom")
            label = int(text.strip("%()"))
            if label in open_rings:
                other, sym = open_rings.pop(label)
                if (sym is not None and pending is not None and BOND_TYPE[sym] is not None and BOND_TYPE[pending] is not None
                        and BOND_TYPE[sym] != BOND_TYPE[pending]):
                    raise SmilesError("conflicting ring-closure bond symbols")
                if other == prev or any({a, b} == {other, prev} for a, b, _ in bonds):
                    raise SmilesError("ring closure duplicates a bond")
                t_close = BOND_TYPE.get(pending) if pending is not None else None
                if pending in ("/", "\\") or sym in ("/", "\\"):
                    directional.add(len(bonds))
                bonds.append([other, prev, t_close if t_close is not None else BOND_TYPE.get(sym)])
            else:
                open_rings[label] = (prev, pending)
            pending = None
        elif kind == "open":
            if prev is None or pending is not None:
                raise SmilesError("misplaced '('")
            stack.append(prev)
        elif kind == "close":
            if not stack or pending is not None:
                raise SmilesError("misplaced ')'")
            prev = stack.pop()
        else:  # dot
            if pending is not None:
                raise SmilesError("bond symbol before '.'")
            prev = None
    if pending is not None or stack or open_rings:
        raise SmilesError("unterminated bond, branch or ring")
    return atoms, bonds, directional


def _connected_without(n, bonds, skip, src, dst):
    adj = [[] for _ in range(n)]
    for k, (a, b, _) in enumerate(bonds):
        if k != skip:
            adj[a].append(b)
            adj[b].append(a)
    seen, todo = {src}, [src]
    while todo:
        u = todo.pop()
        if u == dst:
            return True
        for v in adj[u]:
            if v not in seen:
                seen.add(v)
                todo.append(v)
    return False


def _clean_up(atoms, bonds):
    """RDKit's cleanUp step (RDKit Book, "Sanitization"), in place: N(=O)=O -> [N+]([O-])=O, N=N#N -> N=[N+]=[N-],
    C=P(=O)X -> C=[P+]([O-])X, O=Cl(=O)O -> [O-][Cl+2]([O-])O.  Atoms it touches keep the hydrogens they were written with."""
    def around(i):
        return [(b if a == i else a, k) for k, (a, b, _) in enumerate(bonds) if i in (a, b)]

    def valence(i):
        return int(np.floor(sum(1.5 if bonds[k][2] == 12 else bonds[k][2] for _, k in around(i)) + 0.1 + 0.5)) + atoms[i]["h_explicit"]

    def separate(i, j, k, new_type):
        bonds[k][2] = new_type
        atoms[i]["charge"] += 1
        atoms[j]["charge"] = -1
        atoms[i]["bracket"] = atoms[j]["bracket"] = True

    for i, at in enumerate(atoms):
        if at["charge"] != 0:
            continue
        nbrs = around(i)
        if at["z"] == 7 and valence(i) == 5:
            for j, k in nbrs:
                if atoms[j]["charge"] == 0 and (atoms[j]["z"], bonds[k][2]) in ((8, 2), (7, 3)):
                    separate(i, j, k, bonds[k][2] - 1)
                    break
        elif at["z"] == 15 and valence(i) == 5 and len(nbrs) == 3:
            oxo = [(j, k) for j, k in nbrs if atoms[j]["z"] == 8 and atoms[j]["charge"] == 0 and bonds[k][2] == 2]
            if oxo and any(atoms[j]["z"] in (6, 15) and bonds[k][2] == 2 for j, k in nbrs):
                separate(i, oxo[-1][0], oxo[-1][1], 1)
        elif at["z"] in (17, 35, 53) and valence(i) in (3, 5, 7) and all(atoms[j]["z"] == 8 for j, _ in nbrs):
            for j, k in nbrs:
                if bonds[k][2] == 2:
                    separate(i, j, k, 1)


def molecule(smiles: str):
    """SMILES -> (atom table (n, 6) int [Z, charge, isotope, total Hs, aromatic, in ring], bond table (m, 4) int
    [begin, end, RDKit bond type, in ring]) with the rules listed in nvmolkit_amd/csrc/smiles.cpp's header."""
    atoms, bonds, directional = parse_with_directions(smiles.split()[0] if smiles.split() else "")
    # fold plain hydrogen atoms into their neighbour (RDKit's default removeHs: not next to a dummy atom, and not when the
    # hydrogen defines double-bond stereo, i.e. its bond carries a direction and the neighbour a double bond)
    degree = [0] * len(atoms)
    for a, b, _ in bonds:
        degree[a] += 1
        degree[b] += 1
    drop = set()
    folded = [0] * len(atoms)
    for i, at in enumerate(atoms):
        if at["z"] == 1 and at["isotope"] == 0 and at["charge"] == 0 and at["h_explicit"] == 0 and degree[i] == 1:
            (k, (a, b, t)), = [(k, bd) for k, bd in enumerate(bonds) if i in bd[:2]]
            other = b if a == i else a
            stereo = k in directional and any(other in bd[:2] and bd[2] == 2 for bd in bonds)
            if atoms[other]["z"] not in (0, 1) and t in (None, 1) and not stereo:
                drop.add(i)
                folded[other] += 1
                if atoms[other]["bracket"] or atoms[other]["aromatic"]:  # '[H]n1cccc1' is [nH]
                    atoms[other]["h_explicit"] += 1
                    atoms[other]["bracket"] = True
    # an organic-subset atom drawn with its hydrogens in one of its higher valence states keeps them (RDKit's removeHs:
    # H3P=O stays H3P=O; a recount from the other bonds would make it HP=O); all others are recounted further down
    for i, at in enumerate(atoms):
        if folded[i] and not at["bracket"] and not at["aromatic"] and at["z"] in VALENCES:
            drawn = sum(1.5 if t == 12 else float(t or 1) for a, b, t in bonds if i in (a, b))
            state = next((v for v in VALENCES[at["z"]] if v >= int(np.floor(drawn + 0.1 + 0.5))), None)
            if state in VALENCES[at["z"]][1:]:
                at["kept_h"] = folded[i]
    keep = [i for i in range(len(atoms)) if i not in drop]
    renum = {old: new for new, old in enumerate(keep)}
    atoms = [atoms[i] for i in keep]
    bonds = [[renum[a], renum[b], t] for a, b, t in bonds if a not in drop and b not in drop]
    n = len(atoms)
    in_ring_bond = [_connected_without(n, bonds, k, a, b) for k, (a, b, _) in enumerate(bonds)]
    for k, bd in enumerate(bonds):
        if bd[2] is None:
            bd[2] = 12 if (atoms[bd[0]]["aromatic"] and atoms[bd[1]]["aromatic"] and in_ring_bond[k]) else 1
    _clean_up(atoms, bonds)
    # implicit hydrogens of organic-subset atoms
    order_sum = [0.0] * n
    for a, b, t in bonds:
        w = 1.5 if t == 12 else float(t)
        order_sum[a] += w
        order_sum[b] += w
    total_h = []
    for i, at in enumerate(atoms):
        if at["bracket"] or at["z"] == 0:
            total_h.append(at["h_explicit"])
            # RDKit's strict valence check on a bracket atom of B / C / N / O: at most the element's highest valence, one more
            # per positive and one fewer per negative charge (boron: the other way round; a carbocation loses one too)
            top = {5: 3, 6: 4, 7: 3, 8: 2}.get(at["z"])
            if at["bracket"] and top is not None and not at["aromatic"]:
                q = at["charge"]
                top += -q if at["z"] == 5 else (-abs(q) if at["z"] == 6 else q)
                if int(np.floor(order_sum[i] + 0.1 + 0.5)) + at["h_explicit"] > top:
                    raise SmilesError(f"valence of bracket atom {i} is not allowed")
            continue
        allowed = VALENCES[at["z"]]
        acc = order_sum[i]
        if at["aromatic"]:
            default = allowed[0]
            if acc > default:
                acc = max([v for v in allowed if v <= acc] or [default])
            ev = int(np.floor(acc + 0.1 + 0.5))
            total_h.append(max(default - ev, 0))
        else:
            kept = at.get("kept_h", 0)
            ev = int(np.floor(acc + 0.1 + 0.5)) + kept
            fits = [v for v in allowed if v >= ev]
            if not fits:
                raise SmilesError(f"valence {ev} of atom {i} is not allowed")
            total_h.append(kept + fits[0] - ev)
    ring_atom = [False] * n
    for k, (a, b, _) in enumerate(bonds):
        if in_ring_bond[k]:
            ring_atom[a] = ring_atom[b] = True
    atom_table = np.array([[at["z"], at["charge"], at["isotope"], total_h[i], int(at["aromatic"]), int(ring_atom[i])]
                           for i, at in enumerate(atoms)], dtype=np.int64).reshape(n, 6)
    bond_table = np.array([[a, b, t, int(in_ring_bond[k])] for k, (a, b, t) in enumerate(bonds)], dtype=np.int64).reshape(len(bonds), 4)
    return atom_table, bond_table


def invariant_components(atom_table, bond_table):
    """(n, 5) components [Z, degree + Hs, Hs incl. hydrogen neighbours, charge, int(mass - average mass)] + ring flags."""
    n = len(atom_table)
    degree = np.zeros(n, dtype=np.int64)
    nbr_h = np.zeros(n, dtype=np.int64)
    for a, b, _, _ in bond_table:
        degree[a] += 1
        degree[b] += 1
        nbr_h[a] += atom_table[b, 0] == 1
        nbr_h[b] += atom_table[a, 0] == 1
    dmass = np.zeros(n, dtype=np.int64)
    for i, (z, _, iso, *_rest) in enumerate(atom_table):
        if iso:
            dmass[i] = int(ISOTOPE.get((int(z), int(iso)), float(iso)) - WEIGHT[int(z)])
    comps = np.stack([atom_table[:, 0], atom_table[:, 3] + degree, atom_table[:, 3] + nbr_h, atom_table[:, 1], dmass], 1)
    return comps, atom_table[:, 5].astype(bool)


def self_matches(atom_table, bond_table, symmetrize_terminal_groups: bool = True):
    """Every mapping of the (hydrogen-free) graph onto itself that keeps element, charge, isotope and each bond with its type, as
    a sorted list of tuples: image[i] of atom i.  By exhaustion over the permutations within each class of like atoms, so only
    for small molecules; the count and the set are what RDKit's SubstructMatch(mol, mol, uniquify=False) returns
    (rdkit_extensions/conformer_pruning.cpp:24-60).  ``symmetrize_terminal_groups``: the two terminal N / O atoms of a
    conjugated X-A=Y group (X, Y one-coordinate N or O) are made alike first, as RDKit's MolAlign::details::symmetrizeTerminalAtoms
    does for the query molecule: charges ignored, both bonds "single or double"."""
    import itertools

    atoms = np.array(atom_table, dtype=np.int64).reshape(-1, 6)
    bonds = {}
    for a, b, t, _ in np.asarray(bond_table, dtype=np.int64).reshape(-1, 4):
        bonds[(int(a), int(b))] = bonds[(int(b), int(a))] = int(t)
    n = len(atoms)
    degree = [sum(1 for (a, _b) in bonds if a == i) for i in range(n)]
    if symmetrize_terminal_groups:
        read = dict(bonds)
        for c in range(n):
            ends = [x for x in range(n) if (c, x) in read and degree[x] == 1 and atoms[x, 0] in (7, 8)]
            for x in ends:
                for y in ends:
                    if x != y and read[(c, x)] == 1 and read[(c, y)] == 2:
                        for t in (x, y):
                            bonds[(c, t)] = bonds[(t, c)] = -1        # "single or double": a class only such bonds are in
                            atoms[t, 1] = 0
    classes = {}
    for i in range(n):
        classes.setdefault((int(atoms[i, 0]), int(atoms[i, 1]), int(atoms[i, 2]), degree[i]), []).append(i)
    groups = list(classes.values())
    found = []
    for perms in itertools.product(*(itertools.permutations(g) for g in groups)):
        image = [0] * n
        for g, p in zip(groups, perms):
            for i, t in zip(g, p):
                image[i] = t
        if all(bonds.get((image[a], image[b])) == t for (a, b), t in bonds.items()):
            found.append(tuple(image))
    return sorted(found)
