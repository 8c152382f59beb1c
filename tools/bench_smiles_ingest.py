#!/usr/bin/env python3
"""Throughput of the RDKit-free fingerprint ingestion: SMILES -> graphs -> Morgan inputs on the host (all threads / one
thread), and SMILES -> fingerprints end to end on the GPU, on the reference's 10 000 benchmark SMILES repeated --repeat times.
Usage: python tools/bench_smiles_ingest.py [--repeat 20]"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nvmolkit_amd.fingerprints import MorganFingerprintGenerator, SmilesSet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--repeat", type=int, default=20)
args = ap.parse_args()
smiles = [line.split()[0] for line in (ROOT / "tests" / "golden" / "chembl_10k.smi").read_text().splitlines() if line.strip()] * args.repeat
out = {"molecules": len(smiles)}
SmilesSet(smiles[:100])  # library load
for label, threads in (("all_threads", 0), ("one_thread", 1)):
    t_parse = 1e9
    for _ in range(3):  # the first large call of a process also pays for growing the heap; the best of three is reported
        t = time.perf_counter()
        mols = SmilesSet(smiles, threads)
        t_parse = min(t_parse, time.perf_counter() - t)
    size = np.maximum(mols.n_atoms, mols.n_bonds)
    t = time.perf_counter()
    lo = 0
    for b in (32, 64, 128, 256, 512, 1024):
        idx = np.flatnonzero((size >= lo) & (size < b))
        lo = b
        mols.morgan_inputs(idx, b, threads)
    t_inputs = time.perf_counter() - t
    text = ("\n".join(smiles) + "\n").encode()
    t_text = 1e9
    for _ in range(3):
        t = time.perf_counter()
        SmilesSet.from_text(text, threads)
        t_text = min(t_text, time.perf_counter() - t)
    out[label] = {"parse_s": t_parse, "parse_text_buffer_s": t_text, "morgan_inputs_s": t_inputs,
                  "molecules_per_s": len(smiles) / (t_parse + t_inputs)}
if torch.cuda.is_available():
    gen = MorganFingerprintGenerator(2, 2048)
    gen.GetFingerprintsFromSmiles(smiles[:1000]).torch()
    torch.cuda.synchronize()
    t = time.perf_counter()
    fp = gen.GetFingerprintsFromSmiles(smiles).torch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    out["smiles_to_fingerprints_on_gpu"] = {"seconds": dt, "molecules_per_s": len(smiles) / dt,
                                            "bucket_counts": {str(b): int(((size >= lo_) & (size < b)).sum())
                                                              for lo_, b in ((0, 32), (32, 64), (64, 128), (128, 256), (256, 512), (512, 1024))}}
print(json.dumps(out))
