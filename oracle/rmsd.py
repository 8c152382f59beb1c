"""CPU oracle for conformer RMSD matrices and RMS pruning — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy restatement of the reference's math (src/conformer_rmsd.cu:133-258: centre, cross-covariance H, singular values,
RMSD^2 = (Sp + Sq - 2 (s0 + s1 + sgn(det H) s2)) / N; prealigned = plain RMSD of the raw coordinates) with LAPACK's SVD in
place of the closed-form eigenvalues, and of the greedy pruning loop (rdkit_extensions/conformer_pruning.cpp:88-137).
Pinned by closed forms in tests/test_oracle_rmsd.py: rigid motions give 0, a mirror image does not, known displacements."""

from __future__ import annotations

import numpy as np


def pair_rmsd(a: np.ndarray, b: np.ndarray, prealigned: bool = False) -> float:
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    n = len(a)
    if prealigned:
        return float(np.sqrt(((a - b) ** 2).sum() / n))
    p, q = a - a.mean(0), b - b.mean(0)
    h = p.T @ q
    s = np.linalg.svd(h, compute_uv=False)
    if np.linalg.det(h) < 0.0:
        s[2] = -s[2]
    return float(np.sqrt(max(((p * p).sum() + (q * q).sum() - 2.0 * s.sum()) / n, 0.0)))


def rms_matrix(confs: np.ndarray, prealigned: bool = False) -> np.ndarray:
    """Condensed lower triangle: pair (i, j), i > j, at i (i - 1) / 2 + j (RDKit GetConformerRMSMatrix order)."""
    n = len(confs)
    out = np.zeros(n * (n - 1) // 2)
    for i in range(1, n):
        for j in range(i):
            out[i * (i - 1) // 2 + j] = pair_rmsd(confs[i], confs[j], prealigned)
    return out


def prune(confs: np.ndarray, threshold: float) -> np.ndarray:
    """Boolean keep mask of the greedy pruning."""
    keep = np.zeros(len(confs), dtype=bool)
    for i in range(len(confs)):
        keep[i] = all(pair_rmsd(confs[i], confs[k]) >= threshold for k in range(i) if keep[k])
    return keep


def pair_rmsd_sym(a: np.ndarray, b: np.ndarray, matches: np.ndarray) -> float:
    """Smallest superposed RMSD of a[matches[0]] against b[matches[k]] over the self matches k — what the reference's
    _isConfFarFromRest compares with the threshold (rdkit_extensions/conformer_pruning.cpp:88-114: reference points from
    selfMatches[0], probe points from each match, AlignPoints' sum of squares)."""
    matches = np.asarray(matches, dtype=np.int64)
    ref = np.asarray(a, dtype=np.float64)[matches[0]]
    return min(pair_rmsd(ref, np.asarray(b, dtype=np.float64)[mt]) for mt in matches)


def rms_matrix_sym(confs: np.ndarray, matches: np.ndarray) -> np.ndarray:
    """Condensed lower triangle of pair_rmsd_sym: entry i (i - 1) / 2 + j = conformer i as the reference, conformer j probed."""
    n = len(confs)
    out = np.zeros(n * (n - 1) // 2)
    for i in range(1, n):
        for j in range(i):
            out[i * (i - 1) // 2 + j] = pair_rmsd_sym(confs[i], confs[j], matches)
    return out


def prune_sym(confs: np.ndarray, threshold: float, matches: np.ndarray) -> np.ndarray:
    """Greedy pruning with symmetry (addConformersToMoleculeWithPruning with useSymmetryForPruning)."""
    keep = np.zeros(len(confs), dtype=bool)
    for i in range(len(confs)):
        keep[i] = all(pair_rmsd_sym(confs[i], confs[k], matches) >= threshold for k in range(i) if keep[k])
    return keep
