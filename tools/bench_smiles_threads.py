#!/usr/bin/env python3
"""Thread scaling of the host side of the RDKit-free ingestion: SMILES text -> graphs (nvmk_smiles_parse_text) and graphs ->
Morgan kernel inputs (nvmk_smiles_morgan_inputs) on 1, 2, 4, ... threads.  Usage: python tools/bench_smiles_threads.py [--repeat 100]"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

from nvmolkit_amd.fingerprints import SmilesSet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--repeat", type=int, default=100)
args = ap.parse_args()
smiles = [line.split()[0] for line in (ROOT / "tests" / "golden" / "chembl_10k.smi").read_text().splitlines() if line.strip()] * args.repeat
text = ("\n".join(smiles) + "\n").encode()
SmilesSet.from_text(text[:200000], 1)
out = {"molecules": len(smiles), "host_threads": os.cpu_count(), "parse_text_s": {}, "morgan_inputs_s": {}}
threads = [t for t in (1, 2, 4, 8, 16, 32, 64, 128, 256) if t <= (os.cpu_count() or 1)]
for thr in threads:
    best = 1e9
    mols = None
    for _ in range(3 if thr > 1 else 1):
        del mols  # (freeing the previous set — 0.6 GB at 1M molecules — is not part of parsing the next one)
        t = time.perf_counter()
        mols = SmilesSet.from_text(text, thr)
        best = min(best, time.perf_counter() - t)
    out["parse_text_s"][str(thr)] = best
    size = np.maximum(mols.n_atoms, mols.n_bonds)
    idx = np.flatnonzero(size < 64)
    best = 1e9
    for _ in range(2):
        t = time.perf_counter()
        mols.morgan_inputs(idx, 64, thr)
        best = min(best, time.perf_counter() - t)
    out["morgan_inputs_s"][str(thr)] = best
    del mols
print(json.dumps(out))
