import sys, json
sys.path.insert(0, '.')
import numpy as np, torch
from nvmolkit_amd import synthetic
from nvmolkit_amd.forcefield import DG, ETK, MMFF, UFF, FlatForcefieldBatch
from oracle import ffc
W = {DG: (0.7, 0.3), ETK: (1.0, 1.0), MMFF: (1.0, 1.0), UFF: (1.0, 1.0)}
SIZES = [5, 12, 24, 48, 64, 96, 150, 200]
for kind in (DG, ETK, MMFF, UFF):
    rng = np.random.default_rng(500 + kind)
    systems = [synthetic.random_ff_system(kind, n, rng) for n in SIZES]
    a_s, flat, groups = synthetic.build_ff_batch_arrays(kind, systems)
    gpu = FlatForcefieldBatch(kind, a_s, groups); cpu = ffc.Batch(kind, a_s, groups)
    w0, w1 = W[kind]
    for iters in (10, 30, 60, 100):
        pos = torch.from_numpy(flat).cuda()
        e, st, it = gpu.minimize(pos, max_iters=iters, grad_tol=1e-14, w0=w0, w1=w1)
        x, ec, stc, itc = cpu.minimize(flat, max_iters=iters, grad_tol=1e-14, w0=w0, w1=w1)
        got = pos.cpu().numpy(); dim = gpu.dim
        dev = [float(np.max(np.abs(got[a_s[s]*dim:a_s[s+1]*dim] - x[a_s[s]*dim:a_s[s+1]*dim]))) for s in range(len(SIZES))]
        print(json.dumps({"kind": int(kind), "iters": iters, "same_iters": bool(np.array_equal(it.cpu().numpy(), itc)), "max_dev_per_system": ["%.1e" % d for d in dev],
                          "rel_energy_dev": "%.1e" % float(np.max(np.abs(e.cpu().numpy() - ec) / np.maximum(1.0, np.abs(ec))))}), flush=True)
