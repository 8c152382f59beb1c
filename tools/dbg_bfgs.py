import sys, numpy as np, torch
sys.path.insert(0, '.')
from nvmolkit_amd.forcefield import DG, ETK, MMFF, UFF, FlatForcefieldBatch
from tests import util
kind = int(sys.argv[1]); sizes=[int(x) for x in sys.argv[2:]]
rng = np.random.default_rng(kind + 21)
systems = [util.random_ff_system(kind, n, rng) for n in sizes]
a_s, flat, groups = util.build_ff_batch_arrays(kind, systems)
batch = FlatForcefieldBatch(kind, a_s, groups); pos = torch.from_numpy(flat).cuda()
e0 = batch.compute_energy(pos).cpu().numpy()
e, st, it = batch.minimize(pos, max_iters=300)
torch.cuda.synchronize()
print(kind, sizes, e0, e.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy())
