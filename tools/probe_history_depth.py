#!/usr/bin/env python3
"""How far do the two forms of a team system's inverse Hessian — packed triangle, history of the rank-2 updates — and the CPU
oracle agree, by depth of the minimisation?  Prints one JSON line per (kind, iterations): largest coordinate deviation of either
form from the oracle and from each other, per system.  (Sets the tolerances of tests/test_bfgs_parity_gpu.py's history tests.)"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nvmolkit_amd import _native, synthetic  # noqa: E402
from nvmolkit_amd.forcefield import DG, MMFF, FlatForcefieldBatch  # noqa: E402
from oracle import ffc  # noqa: E402

W = {DG: (0.7, 0.3), MMFF: (1.0, 1.0)}
SIZES = [300, 500]
for kind in (DG, MMFF):
    rng = np.random.default_rng(2500 + kind)
    systems = [synthetic.random_ff_system(kind, n, rng) for n in SIZES]
    a_s, flat, groups = synthetic.build_ff_batch_arrays(kind, systems)
    gpu = FlatForcefieldBatch(kind, a_s, groups)
    cpu = ffc.Batch(kind, a_s, groups)
    w0, w1 = W[kind]
    dim = gpu.dim
    for iters in (10, 30, 60, 120):
        out = {}
        for history in ("0", "auto"):
            with _native.options(NVMK_BFGS_HISTORY=history, NVMK_BFGS_TEAM_TIMEOUT_MS="5000"):
                pos = torch.from_numpy(flat).cuda()
                e, st, it = gpu.minimize(pos, max_iters=iters, grad_tol=1e-14, w0=w0, w1=w1)
                out[history] = (pos.cpu().numpy(), e.cpu().numpy(), it.cpu().numpy())
        x, ec, stc, itc = cpu.minimize(flat, max_iters=iters, grad_tol=1e-14, w0=w0, w1=w1) if iters <= 60 else (None, None, None, None)

        def dev(a, b):
            return ["%.1e" % float(np.max(np.abs(a[a_s[s] * dim:a_s[s + 1] * dim] - b[a_s[s] * dim:a_s[s + 1] * dim]))) for s in range(len(SIZES))]

        print(json.dumps({"kind": int(kind), "iters": iters, "iterations": [out["0"][2].tolist(), out["auto"][2].tolist(), None if itc is None else itc.tolist()],
                          "triangle_vs_history": dev(out["0"][0], out["auto"][0]),
                          "triangle_vs_oracle": None if x is None else dev(out["0"][0], x),
                          "history_vs_oracle": None if x is None else dev(out["auto"][0], x),
                          "energies": [out["0"][1].tolist(), out["auto"][1].tolist(), None if ec is None else ec.tolist()]}), flush=True)
