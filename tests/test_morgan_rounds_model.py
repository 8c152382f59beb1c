"""The dedup-without-sorting of the Morgan kernel (nvmolkit_amd/csrc/morgan.hip) restated in Python and held to the C oracle's
line-by-line restatement of RDKit's sorted sweep (oracle/oracle_morgan.c = src/morgan_fingerprint_cpu.cpp:61-255) — without a GPU.
The kernel's claim: the reference's sort only decides, among atoms whose bond-neighbourhood bitsets are EQUAL, which one comes
first; so an atom's environment survives a round iff its bitset was not accepted in an earlier round and no other live atom has
the same bitset with a smaller (invariant, atom index).  Inputs: the reference's ChEMBL benchmark SMILES through the library's own
host-side ingestion (nvmk_smiles_morgan_inputs), plus graphs with many symmetric atoms (the case the tie-break exists for)."""

from pathlib import Path

import numpy as np
import pytest

import oracle
from nvmolkit_amd.fingerprints import SmilesSet

ROOT = Path(__file__).resolve().parents[1]
M32 = 0xFFFFFFFF


def hash_combine(seed, v):
    return (seed ^ ((v + 0x9E3779B9 + ((seed << 6) & M32) + (seed >> 2)) & M32)) & M32


def morgan_no_sort(atom_inv, bond_inv, bond_idx, bond_other, n, radius, fp_bits):
    """One molecule: the kernel's rounds with Python ints as bitsets (bit b = bond b)."""
    bonds = [[(int(bond_idx[a, k]), int(bond_other[a, k])) for k in range(bond_idx.shape[1]) if bond_idx[a, k] >= 0] for a in range(n)]
    fp = set(int(atom_inv[a]) % fp_bits for a in range(n))
    cur = [int(atom_inv[a]) for a in range(n)]
    nbh = [0] * n
    rn = [0] * n                      # a lane's running neighbourhood (registers: kept across rounds)
    dead = [False] * n
    seen = []
    for layer in range(radius):
        computed, invar = [False] * n, [0] * n
        for a in range(n):
            if dead[a]:
                continue
            if not bonds[a]:
                dead[a] = True        # isolated atom: never produces an environment
                continue
            pairs = []
            for b, o in bonds[a]:
                rn[a] |= nbh[o] | (1 << b)
                pairs.append((int(np.int32(np.uint32(bond_inv[b]))), cur[o]))  # (bond type as int32, neighbour invariant)
            pairs.sort()
            h = hash_combine(layer, cur[a])
            for t, v in pairs:
                h = hash_combine(h, hash_combine(hash_combine(0, t & M32), v))
            computed[a], invar[a] = True, h
        accepted = []
        for a in range(n):
            if not computed[a]:
                continue
            lose = rn[a] in seen
            for b in range(n):
                if not lose and b != a and computed[b] and rn[b] == rn[a]:
                    lose = invar[b] < invar[a] or (invar[b] == invar[a] and b < a)
            dead[a] = lose
            if not lose:
                accepted.append(a)
        for a in accepted:            # appended after everyone has looked at `seen`
            seen.append(rn[a])
            fp.add(invar[a] % fp_bits)
        for a in range(n):            # roll
            cur[a] = invar[a] if computed[a] else 0
            if computed[a]:
                nbh[a] = rn[a]
    out = np.zeros(fp_bits // 32, dtype=np.uint32)
    for bit in fp:
        out[bit >> 5] |= np.uint32(1 << (bit & 31))
    return out


def check(mols, idx, stride, radius, fp_bits):
    atom_inv, bond_inv, bond_idx, bond_other, n_atoms = mols.morgan_inputs(idx, stride)
    want = oracle.morgan_fingerprints(atom_inv, bond_inv, bond_idx, bond_other, n_atoms, stride, radius, fp_bits)
    bix = np.asarray(bond_idx).reshape(len(idx), stride, -1)
    bot = np.asarray(bond_other).reshape(len(idx), stride, -1)
    ai = np.asarray(atom_inv).reshape(len(idx), stride)
    bi = np.asarray(bond_inv).reshape(len(idx), stride)
    for m in range(len(idx)):
        got = morgan_no_sort(ai[m], bi[m], bix[m], bot[m], int(n_atoms[m]), radius, fp_bits)
        assert np.array_equal(got, want[m]), (int(idx[m]), radius)


@pytest.mark.parametrize("radius,fp_bits", [(0, 2048), (1, 1024), (2, 2048), (3, 2048), (4, 512)])
def test_no_sort_rounds_equal_the_sorted_sweep_on_chembl(radius, fp_bits):
    smiles = [line.split()[0] for line in (ROOT / "tests" / "golden" / "chembl_10k.smi").read_text().splitlines() if line.strip()]
    mols = SmilesSet(smiles[::67][:150])
    size = np.maximum(mols.n_atoms, mols.n_bonds)
    idx = np.flatnonzero((mols.status == 0) & (size < 64))
    assert len(idx) > 100
    check(mols, idx, 64, radius, fp_bits)


def test_symmetric_molecules():
    # many atoms with identical environments: the (invariant, index) tie-break decides which copy survives
    mols = SmilesSet(["C1CCCCC1", "c1ccccc1", "CC(C)(C)C", "C1CC1", "C12C3C4C1C5C2C3C45", "FC(F)(F)C(F)(F)F", "CCCCCCCCCCCC",
                      "c1ccc2ccccc2c1", "C1CCC2(CC1)CCCCC2", "OCC(CO)(CO)CO", "C", "CC", "[Na+].[Cl-]", "C=C", "C#C"])
    idx = np.flatnonzero(mols.status == 0)
    assert len(idx) == 15
    for radius in (0, 1, 2, 3, 5):
        check(mols, idx, 32, radius, 2048)
