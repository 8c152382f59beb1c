"""TEST INFRASTRUCTURE — an independent reader of MDL molfile (V2000) records, for the SD-file ingestion tests.

The product reads SD files in C++ (nvmolkit_amd/csrc/smiles.cpp: MolfileReader, fixed columns).  This restatement splits
lines on white space where the format allows it and keeps every atom (hydrogens included), so that the tests can compare
what the product derives — folded hydrogens, hydrogen counts from the valence model, charges, bond types — with what the
file spells out.  Only tests/ may import this module."""

from __future__ import annotations

import numpy as np

from oracle.smiles import Z_OF

CHARGE_CODE = {0: 0, 1: 3, 2: 2, 3: 1, 5: -1, 6: -2, 7: -3}


def records(text: str):
    """the molfile records of an SD file's text (data items dropped)"""
    out, cur = [], []
    for line in text.splitlines():
        if line.startswith("$$$$"):
            out.append(cur)
            cur = []
        else:
            cur.append(line)
    if any(l.strip() for l in cur):
        out.append(cur)
    return out


def read(lines):
    """-> (name, Z (n,), charge (n,), isotope (n,), bonds (m, 3) [a, b, type]) with every atom of the record"""
    name = lines[0].strip()
    n_atoms, n_bonds = int(lines[3][0:3]), int(lines[3][3:6])
    z, charge, isotope = [], [], []
    for line in lines[4:4 + n_atoms]:
        parts = line.split()
        z.append(Z_OF[parts[3]])
        charge.append(CHARGE_CODE[int(line[36:39] or 0)])
        isotope.append(0)
    bonds = [[int(l[0:3]) - 1, int(l[3:6]) - 1, int(l[6:9])] for l in lines[4 + n_atoms:4 + n_atoms + n_bonds]]
    first_chg = True
    for line in lines[4 + n_atoms + n_bonds:]:
        if line.startswith("M  END"):
            break
        if line.startswith("M  CHG") or line.startswith("M  ISO"):
            if line.startswith("M  CHG") and first_chg:
                charge, first_chg = [0] * n_atoms, False
            nums = [int(x) for x in line[6:].split()]
            for a, v in zip(nums[1::2], nums[2::2]):
                (charge if line[3:6] == "CHG" else isotope)[a - 1] = v
    return name, np.array(z), np.array(charge), np.array(isotope), np.array(bonds, dtype=np.int64).reshape(-1, 3)


def heavy_atom_graph(z, charge, isotope, bonds):
    """What RDKit's removeHs leaves: plain hydrogens (no isotope, no charge, one single bond to a heavier atom) are dropped and
    counted on their neighbour.  -> (atoms (k, 4) [Z, charge, isotope, hydrogens that were drawn], bonds (j, 3), kept indices)"""
    n = len(z)
    degree = np.zeros(n, dtype=int)
    for a, b, _ in bonds:
        degree[a] += 1
        degree[b] += 1
    drop = np.zeros(n, dtype=bool)
    drawn_h = np.zeros(n, dtype=int)
    for a, b, t in bonds:
        for h, other in ((a, b), (b, a)):
            if z[h] == 1 and isotope[h] == 0 and charge[h] == 0 and degree[h] == 1 and z[other] != 1 and t == 1:
                drop[h] = True
                drawn_h[other] += 1
    keep = np.flatnonzero(~drop)
    renum = -np.ones(n, dtype=int)
    renum[keep] = np.arange(len(keep))
    atoms = np.stack([z[keep], charge[keep], isotope[keep], drawn_h[keep]], 1)
    kept_bonds = np.array([[renum[a], renum[b], t] for a, b, t in bonds if not drop[a] and not drop[b]], dtype=np.int64).reshape(-1, 3)
    return atoms, kept_bonds, keep


def write_molblock(atom_table, bond_types, bond_table, name="oracle") -> str:
    """A V2000 record of a molecule given as tables (oracle.smiles.molecule): heavy atoms only — hydrogen counts cannot be
    written in a molfile and are left to the reader's valence model — charges, isotopes and radicals as M  CHG / M  ISO /
    M  RAD lines, bond types
    1 / 2 / 3 (pass a Kekule assignment for aromatic molecules: oracle.aromaticity.kekulize)."""
    from oracle.smiles import ELEMENTS

    lines = [name, "  oracle", "", f"{len(atom_table):3d}{len(bond_table):3d}  0  0  0  0  0  0  0  0999 V2000"]
    for row in atom_table:
        lines.append(f"{0.0:10.4f}{0.0:10.4f}{0.0:10.4f} {ELEMENTS[int(row[0])]:<3s} 0  0  0  0  0  0  0  0  0  0  0  0")
    for (a, b, _, _), t in zip(bond_table, bond_types):
        lines.append(f"{int(a) + 1:3d}{int(b) + 1:3d}{int(t):3d}  0")
    # an atom whose bonds and hydrogens stop short of its (charge-shifted) valence is a radical: M  RAD, 2 = doublet, 3 = triplet
    from oracle.smiles import VALENCES

    bonded = np.zeros(len(atom_table))
    for (a, b, _, _), t in zip(bond_table, bond_types):
        bonded[a] += t
        bonded[b] += t
    radicals = []
    for i, row in enumerate(atom_table):
        z, q, h = int(row[0]), int(row[1]), int(row[3])
        if z in VALENCES:
            shift = -q if z == 5 else (-abs(q) if z == 6 else q)
            state = next((v + shift for v in VALENCES[z] if v + shift >= bonded[i] + h), None)
            if state is not None and state - bonded[i] - h > 0:
                radicals.append((i + 1, 2 if state - bonded[i] - h == 1 else 3))
    for tag, col in (("CHG", 1), ("ISO", 2), ("RAD", None)):
        entries = radicals if col is None else [(i + 1, int(row[col])) for i, row in enumerate(atom_table) if int(row[col]) != 0]
        for lo in range(0, len(entries), 8):
            part = entries[lo:lo + 8]
            lines.append(f"M  {tag}{len(part):3d}" + "".join(f"{a:4d}{v:4d}" for a, v in part))
    lines.append("M  END")
    return "\n".join(lines) + "\n$$$$\n"
