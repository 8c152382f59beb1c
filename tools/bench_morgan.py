#!/usr/bin/env python3
"""Kernel throughput of Morgan fingerprints from flattened invariants (radius 2, 2048 bits) on device-resident inputs:
20k synthetic molecules (tests/util.random_molecule_batch, < 64 atoms) tiled to --mols molecules on the GPU.
Usage: python tools/bench_morgan.py [--mols 1000000] [--stride 64]"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nvmolkit_amd import _native  # noqa: E402
from tests import util  # noqa: E402  (synthetic molecule generator only)

ap = argparse.ArgumentParser()
ap.add_argument("--mols", type=int, default=1_000_000)
ap.add_argument("--stride", type=int, default=64)
ap.add_argument("--fp-bits", type=int, default=2048)
args = ap.parse_args()
base = util.random_molecule_batch(20_000, args.stride, seed=7, min_atoms=8)
flat = util.flatten_molecules(base, args.stride)
reps = (args.mols + len(base) - 1) // len(base)
dev = torch.device("cuda", 0)
d = [torch.from_numpy(np.ascontiguousarray(a.view(np.int32) if a.dtype == np.uint32 else a)).to(dev) for a in flat]
d = [t.repeat((reps,) + (1,) * (t.dim() - 1))[: args.mols].contiguous() for t in d]
out = torch.empty((args.mols, args.fp_bits // 32), dtype=torch.int32, device=dev)
lib, sptr = _native.lib(), _native.stream_ptr(None)
call = lambda: _native.check(lib.nvmk_morgan_from_invariants(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),  # noqa: E731
                                                            d[4].data_ptr(), None, args.mols, args.stride, 2, args.fp_bits,
                                                            out.data_ptr(), sptr))
call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
in_bytes = sum(t.numel() * t.element_size() for t in d)
print(json.dumps({"mols": args.mols, "stride": args.stride, "fp_bits": args.fp_bits, "seconds": dt, "mols_per_s": args.mols / dt,
                  "mean_atoms": float(d[4].float().mean()), "GB_per_s_in_plus_out": (in_bytes + out.numel() * 4) / dt / 1e9}))
