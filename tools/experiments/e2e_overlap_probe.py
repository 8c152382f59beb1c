#!/usr/bin/env python3
"""Where do the 3.5 s go that the whole-file job takes end to end beyond its two GPU stages?  Times, on the ChEMBL file (cache of
tools/bench_conformers.py): the MMFF table assembly alone, ETKDG alone (molecule set resident), ETKDG with the MMFF assembly
running beside it, and ETKDG with the molecule set assembled inside the clock."""
import pickle, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch
from nvmolkit_amd import mmffOptimization
from nvmolkit_amd.embedMolecules import FlatMolecule, FlatMoleculeSet, embed_flat
from nvmolkit_amd.types import CoordinateOutput

cache = Path(sys.argv[1])
lib = pickle.load(open(cache, "rb"))
print("molecules", len(lib), flush=True)
def embed(ms):
    torch.cuda.synchronize(); t = time.perf_counter()
    d = embed_flat(ms, confs_per_molecule=10, max_iterations=10, seed=1, output=CoordinateOutput.DEVICE)
    torch.cuda.synchronize(); return time.perf_counter() - t, d
ms = FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in lib])
embed(FlatMoleculeSet([FlatMolecule(**lib[0]["embed"])]))
t, _ = embed(ms); print("etkdg alone (set resident)          %.2f s" % t, flush=True)
t0 = time.perf_counter(); tb = mmffOptimization.resident_tables([m["mmff"] for m in lib]); torch.cuda.synchronize()
print("mmff tables alone                   %.2f s" % (time.perf_counter() - t0), flush=True); del tb
t0 = time.perf_counter(); pend = mmffOptimization.resident_tables([m["mmff"] for m in lib], wait=False)
t, _ = embed(ms); t1 = time.perf_counter(); tb = pend.result(); torch.cuda.synchronize()
print("etkdg beside the mmff assembly      %.2f s (+ %.2f s waiting for the tables)" % (t, time.perf_counter() - t1), flush=True); del tb
t0 = time.perf_counter(); ms2 = FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in lib]); t, _ = embed(ms2)
print("set assembly + etkdg                %.2f s" % (time.perf_counter() - t0), flush=True)
t, _ = embed(ms); print("etkdg alone again                   %.2f s" % t, flush=True)

# as bench.py runs the job: the molecule set's asynchronous fill AND the MMFF assembly both start before the first ETKDG batch
for label, use_after in (("set + mmff assembly started together + etkdg", False), ("the same, mmff assembly after the set's fill", True)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ms3 = FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in lib])
    pend = mmffOptimization.resident_tables([m["mmff"] for m in lib], wait=False, after=ms3 if use_after else None)
    d = embed_flat(ms3, confs_per_molecule=10, max_iterations=10, seed=1, output=CoordinateOutput.DEVICE)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    tb = pend.result(); torch.cuda.synchronize()
    print("%-50s %.2f s (+ %.2f s waiting for the tables)" % (label, t1 - t0, time.perf_counter() - t1), flush=True)
    del tb, ms3, d
