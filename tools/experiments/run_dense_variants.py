#!/usr/bin/env python3
"""Times the rejected dense / count kernel variants of tools/experiments/fp4_experiments.hip against each other on the GPU
(needs tools/experiments/build.sh first).  Usage: python tools/experiments/run_dense_variants.py [rows] [refs]"""
import ctypes
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from bench import SEED, synth_fingerprints  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
refs = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
L = ctypes.CDLL(str(Path(__file__).resolve().parent / "libnvmk_experiments.so"))
L.nvmkx_fp4_workspace_bytes.restype = ctypes.c_size_t
L.nvmkx_fp4_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int]
vp, i64 = ctypes.c_void_p, ctypes.c_int64
L.nvmkx_fp4_prepare.argtypes = [vp, i64, ctypes.c_int, vp, vp]
L.nvmkx_cross_similarity_prepared_f64.argtypes = [ctypes.c_int, vp, i64, i64, i64, vp, i64, ctypes.c_int, vp, i64, vp]
dev = torch.device("cuda")
q = synth_fingerprints(rows, 64, dev, SEED + 1)
r = synth_fingerprints(refs, 64, dev, SEED)
wq = torch.empty(L.nvmkx_fp4_workspace_bytes(rows, 2048), dtype=torch.uint8, device=dev)
wr = torch.empty(L.nvmkx_fp4_workspace_bytes(refs, 2048), dtype=torch.uint8, device=dev)
out = torch.empty((rows, refs), dtype=torch.float64, device=dev)
s = int(torch.cuda.current_stream().cuda_stream)
L.nvmkx_fp4_prepare(q.data_ptr(), rows, 2048, wq.data_ptr(), s)
L.nvmkx_fp4_prepare(r.data_ptr(), refs, 2048, wr.data_ptr(), s)
for variant in ("", "pipe", "wide", "pp"):
    os.environ["NVMK_DENSE_KERNEL"] = variant
    for _ in range(2):
        L.nvmkx_cross_similarity_prepared_f64(0, wq.data_ptr(), rows, 0, rows, wr.data_ptr(), refs, 2048, out.data_ptr(), refs, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        L.nvmkx_cross_similarity_prepared_f64(0, wq.data_ptr(), rows, 0, rows, wr.data_ptr(), refs, 2048, out.data_ptr(), refs, s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"dense variant {variant or 'tile (the shipped kernel)':28s} {rows * refs / dt / 1e12:.3f} T pairs/s")
