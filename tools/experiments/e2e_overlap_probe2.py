#!/usr/bin/env python3
"""Second look at what the whole-file job pays end to end beyond its two GPU stages (round 6, last session): the two table
assemblies alone by thread count, and ETKDG beside them by thread count and by whether the molecule set is filled synchronously.
Usage: e2e_overlap_probe2.py <directory of tools/bench_conformers.py --cache> (the library is generated when the cache is empty)."""
import pickle, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch
from nvmolkit_amd import mmffOptimization, synthetic
from nvmolkit_amd.embedMolecules import FlatMolecule, FlatMoleculeSet, embed_flat
from nvmolkit_amd.types import CoordinateOutput

cache = Path(sys.argv[1]) / "chembl_10000_100000.pkl"
if cache.exists():
    lib = pickle.load(open(cache, "rb"))
else:
    lib, _ = synthetic.smiles_file_library(ROOT / "tests" / "golden" / "chembl_10k.smi", n_mols=10000, max_atoms=100000)
    cache.parent.mkdir(parents=True, exist_ok=True)
    pickle.dump(lib, open(cache, "wb"), protocol=pickle.HIGHEST_PROTOCOL)
print("molecules", len(lib), flush=True)
sync = torch.cuda.synchronize
def embed(ms):
    sync(); t = time.perf_counter()
    d = embed_flat(ms, confs_per_molecule=10, max_iterations=10, seed=1, output=CoordinateOutput.DEVICE)
    sync(); return time.perf_counter() - t, d
def molset(threads=-1, asynchronous=True):
    return FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in lib], preprocessing_threads=threads, asynchronous=asynchronous)
ms = molset()
embed(FlatMoleculeSet([FlatMolecule(**lib[0]["embed"])]))
for _ in range(2):
    t, _ = embed(ms); print("etkdg alone (set resident)                         %.2f s" % t, flush=True)
for thr in (64, 16, 64):
    sync(); t0 = time.perf_counter(); m2 = molset(thr, False); sync()
    print("molecule set alone, synchronous, %2d threads        %.2f s" % (thr, time.perf_counter() - t0), flush=True); del m2
for thr in (64, 16):
    sync(); t0 = time.perf_counter(); tb = mmffOptimization.resident_tables([m["mmff"] for m in lib], preprocessing_threads=thr); sync()
    print("mmff tables alone, %2d threads                      %.2f s" % (thr, time.perf_counter() - t0), flush=True); del tb
def e2e(label, thr_set, thr_mmff, asynchronous, with_mmff=True):
    sync(); t0 = time.perf_counter()
    m3 = molset(thr_set, asynchronous)
    pend = mmffOptimization.resident_tables([m["mmff"] for m in lib], wait=False, after=m3, preprocessing_threads=thr_mmff) if with_mmff else None
    d = embed_flat(m3, confs_per_molecule=10, max_iterations=10, seed=1, output=CoordinateOutput.DEVICE)
    sync(); t1 = time.perf_counter()
    if pend is not None:
        tb = pend.result(); sync()
    print("%-50s %.2f s (+ %.2f s waiting for the tables)" % (label, t1 - t0, time.perf_counter() - t1), flush=True)
for _ in range(2):
    e2e("as bench.py runs it (async set, mmff after, 64 / 64)", -1, -1, True)
    e2e("async set alone, no mmff assembly", -1, -1, True, False)
    e2e("16 threads for both", 16, 16, True)
    e2e("8 threads for the mmff assembly", -1, 8, True)
    e2e("synchronous set, mmff beside etkdg (64 / 64)", -1, -1, False)
    e2e("synchronous set, mmff beside etkdg on 8 threads", -1, 8, False)
    t, _ = embed(ms); print("etkdg alone again                                  %.2f s" % t, flush=True)
