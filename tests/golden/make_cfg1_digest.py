#!/usr/bin/env python3
"""Writes tests/golden/cfg1_chembl_10k_digest.json: digests of BASELINE.json configs[0] on the reference's benchmark molecules
— 10 000 ChEMBL SMILES -> Morgan radius 2 / 2048 bit -> 10 000 x 10 000 Tanimoto — computed entirely on the CPU: the
library's SMILES ingestion (host code), then the oracle's Morgan and similarity.  The Morgan chain is pinned to RDKit by
tests/test_morgan_rdkit_known_answers.py; this file freezes its output on the full workload so that the GPU run, the oracle
and the ingestion are all held to one committed answer.   python tests/golden/make_cfg1_digest.py"""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402
from nvmolkit_amd.fingerprints import SmilesSet  # noqa: E402


def fingerprints(path):
    mols = SmilesSet.from_file(path)
    assert np.all(mols.status == 0)
    size = np.maximum(mols.n_atoms, mols.n_bonds)
    fp = np.zeros((len(mols), 64), dtype=np.uint32)
    lo = 0
    for stride in (32, 64, 128, 256, 512, 1024):
        idx = np.flatnonzero((size >= lo) & (size < stride))
        lo = stride
        if len(idx):
            fp[idx] = oracle.morgan_fingerprints(*mols.morgan_inputs(idx, stride), stride, 2, 2048)
    return fp


def digest(fp):
    hist = np.zeros(101, dtype=np.int64)
    for lo in range(0, len(fp), 1000):
        sim = oracle.cross_similarity(fp[lo:lo + 1000], fp)
        hist += np.bincount(np.floor(sim * 100.0).astype(np.int64).ravel(), minlength=101)
    bits = np.unpackbits(fp.view(np.uint8), axis=1)
    return {"molecules": int(len(fp)), "fingerprints_sha256": hashlib.sha256(np.ascontiguousarray(fp).tobytes()).hexdigest(),
            "bits_set_total": int(bits.sum()), "similarity_histogram_floor_100x": hist.tolist(),
            "butina": {str(cutoff): butina_digest(fp, cutoff) for cutoff in (0.3, 0.6)}}


def butina_digest(fp, cutoff):
    """Butina clustering of the fingerprints at a distance cutoff (0.3 = the similarity threshold 0.7 of BASELINE configs[1]):
    number of clusters, the ten largest sizes, and a hash of (centroid, members) of every cluster in output order."""
    clusters, sizes, centroids = oracle.butina_fused(fp, cutoff)
    flat = np.array([v for c, members in zip(centroids, clusters) for v in (c, len(members), *members)], dtype=np.int64)
    return {"clusters": len(clusters), "largest": [len(c) for c in clusters[:10]], "singletons": sum(len(c) == 1 for c in clusters),
            "sha256": hashlib.sha256(flat.tobytes()).hexdigest()}


if __name__ == "__main__":
    out = digest(fingerprints(ROOT / "tests" / "golden" / "chembl_10k.smi"))
    Path(__file__).with_name("cfg1_chembl_10k_digest.json").write_text(json.dumps(out) + "\n")
    print(json.dumps(out)[:300])
