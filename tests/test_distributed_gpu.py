"""The GPU entry points of the multi-GPU exchange under RCCL (backend "nccl") as a ONE-rank process group on the one-GPU box:
all_gather_rows, fused_butina_sharded (nvmk_butina_pairs + nvmk_butina_from_pairs around one all-reduce and one all-gather) and
merge_device_results — the code bench.py --gpus N and a multi-GPU caller run, with every collective really issued.  The
world-size-2 logic of the exchange is covered on the CPU with gloo (tests/test_distributed_cpu.py); the wire needs more than one
GPU, which this box does not have."""

import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def one_rank_group():
    import torch.distributed as dist

    if dist.is_initialized():  # launched under torch.distributed.run: use its group
        yield dist.group.WORLD
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist.group.WORLD
    dist.destroy_process_group()


def test_all_gather_rows_over_rccl(one_rank_group):
    from nvmolkit_amd.distributed import all_gather_rows, shard_bounds

    x = torch.randint(-2**31, 2**31 - 1, (1003, 64), dtype=torch.int32, device="cuda")
    lo, hi = shard_bounds(1003, 1, 0)
    assert torch.equal(all_gather_rows(x[lo:hi].contiguous(), 1003, group=one_rank_group), x)


def test_sharded_fused_butina_equals_the_single_gpu_call_and_the_oracle(one_rank_group):
    import oracle
    from nvmolkit_amd.clustering import fused_butina
    from nvmolkit_amd.distributed import fused_butina_sharded
    from tests.util import clustered_fingerprints

    fps = clustered_fingerprints(3000, 64, 40)
    x = torch.from_numpy(fps.view(np.int32)).cuda()
    got = fused_butina_sharded(x, 0.35, group=one_rank_group, return_centroids=True)
    assert got == fused_butina(x, 0.35, return_centroids=True) == oracle.butina_fused(fps, 0.35)


def test_merged_device_results_of_one_rank_are_the_results(one_rank_group):
    from nvmolkit_amd import mmffOptimization, synthetic
    from nvmolkit_amd.distributed import merge_device_results, shard_molecules_by_cost
    from nvmolkit_amd.embedMolecules import FlatMolecule, FlatMoleculeSet, embed_flat
    from nvmolkit_amd.types import CoordinateOutput

    lib = synthetic.druglike_library(24, seed=6, mean_atoms=28, processes=1)
    mine = shard_molecules_by_cost(np.array([m["embed"]["n_atoms"] for m in lib]), 1, 0)
    assert sorted(mine.tolist()) == list(range(24))
    sub = [lib[i] for i in mine]
    dev = embed_flat(FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in sub]), confs_per_molecule=3, max_iterations=10, seed=2,
                     output=CoordinateOutput.DEVICE)
    opt = mmffOptimization.optimize_device([m["mmff"] for m in sub], dev, max_iters=50)
    merged = merge_device_results(opt, mine, len(lib), group=one_rank_group)
    assert merged.num_conformers == opt.num_conformers and merged.values.torch().shape == opt.values.torch().shape
    # ordered by (global molecule, conformer): every conformer's coordinates and energy are the local ones
    key = np.lexsort((opt.conf_indices.torch().cpu().numpy(), mine[opt.mol_indices.torch().cpu().numpy()]))
    a_s = opt.atom_starts.torch().cpu().numpy()
    want = np.concatenate([opt.values.torch().cpu().numpy()[a_s[c]:a_s[c + 1]] for c in key])
    assert np.array_equal(merged.values.torch().cpu().numpy(), want)
    assert np.array_equal(merged.energies.torch().cpu().numpy(), opt.energies.torch().cpu().numpy()[key])
