"""World-size-2 gloo test of the multi-GPU sharding logic (runs on CPU): reference shards are all-gathered, every
rank computes its query rows, and the stacked result equals the single-process matrix."""

import os
import socket
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from nvmolkit_amd.distributed import all_gather_rows, shard_bounds
    from tests import util

    ref = util.clustered_fingerprints(203, 8, 7, seed=1)       # identical on every rank
    queries = util.clustered_fingerprints(101, 8, 7, seed=2)
    lo, hi = shard_bounds(len(ref), world, rank)
    gathered = all_gather_rows(torch.from_numpy(ref[lo:hi].view(np.int32).copy()), len(ref))
    assert torch.equal(gathered, torch.from_numpy(ref.view(np.int32)))
    qlo, qhi = shard_bounds(len(queries), world, rank)
    block = oracle.cross_similarity(queries[qlo:qhi], gathered.numpy().view(np.uint32))  # CPU stand-in for the HIP kernel
    np.save(os.path.join(out_dir, f"block{rank}.npy"), block)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_cross_similarity_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, str(ROOT))
    import oracle
    from tests import util

    ref = util.clustered_fingerprints(203, 8, 7, seed=1)
    queries = util.clustered_fingerprints(101, 8, 7, seed=2)
    stacked = np.concatenate([np.load(tmp_path / f"block{r}.npy") for r in range(world)])
    assert np.array_equal(stacked, oracle.cross_similarity(queries, ref))


def test_shard_bounds_cover_everything():
    from nvmolkit_amd.distributed import padded_shard_rows, shard_bounds

    for n in (0, 1, 7, 8, 1_000_003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in spans) <= padded_shard_rows(n, world)
