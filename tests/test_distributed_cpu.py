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


def _merge_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nvmolkit_amd.distributed import merge_device_results, shard_molecules_by_cost
    from nvmolkit_amd.types import Device3DResult

    n_atoms = np.array([5, 9, 3, 7, 4, 8, 6])
    confs = np.array([2, 1, 0, 3, 1, 2, 1])                     # molecule 2 produced nothing
    mine = shard_molecules_by_cost(n_atoms, world, rank)
    # synthetic shard: value of atom a of conformer k of global molecule g is (g, k, a)
    vals, starts, mols, cnfs, energies = [], [0], [], [], []
    for local_m, g in enumerate(mine):
        for k in range(confs[g]):
            for a in range(n_atoms[g]):
                vals.append((float(g), float(k), float(a)))
            starts.append(starts[-1] + int(n_atoms[g]))
            mols.append(local_m)
            cnfs.append(k)
            energies.append(100.0 * g + k)
    local = Device3DResult(torch.tensor(vals, dtype=torch.float64).reshape(-1, 3), torch.tensor(starts, dtype=torch.int32),
                           torch.tensor(mols, dtype=torch.int32), torch.tensor(cnfs, dtype=torch.int32), 0, len(mine),
                           energies=torch.tensor(energies, dtype=torch.float64),
                           converged=torch.ones(len(mols), dtype=torch.int8))
    merged = merge_device_results(local, mine, len(n_atoms))
    torch.save({"values": merged.values.torch(), "starts": merged.atom_starts.torch(), "mols": merged.mol_indices.torch(),
                "confs": merged.conf_indices.torch(), "energies": merged.energies.torch(), "mine": torch.from_numpy(mine)},
               os.path.join(out_dir, f"merged{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_molecule_sharding_and_result_merge(tmp_path):
    """Molecule batches shard with no data-path collective; the per-rank Device3DResult shards merge
    into the single-process order on every rank."""
    from nvmolkit_amd.distributed import shard_molecules_by_cost

    world = 2
    mp.spawn(_merge_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    out = [torch.load(tmp_path / f"merged{r}.pt") for r in range(world)]
    n_atoms = np.array([5, 9, 3, 7, 4, 8, 6])
    confs = np.array([2, 1, 0, 3, 1, 2, 1])
    owned = np.sort(np.concatenate([o["mine"].numpy() for o in out]))
    assert owned.tolist() == list(range(len(n_atoms)))                       # a partition
    loads = [float((n_atoms[o["mine"].numpy()] ** 2).sum()) for o in out]
    assert max(loads) / min(loads) < 1.3                                     # balanced by atoms^2
    for key in ("values", "starts", "mols", "confs", "energies"):
        assert torch.equal(out[0][key], out[1][key])                        # every rank holds the same merged result
    m = out[0]
    want_mols = np.repeat(np.arange(len(n_atoms)), confs)
    assert m["mols"].tolist() == want_mols.tolist()
    assert m["confs"].tolist() == [k for g in range(len(n_atoms)) for k in range(confs[g])]
    assert m["energies"].tolist() == [100.0 * g + k for g in range(len(n_atoms)) for k in range(confs[g])]
    starts = m["starts"].tolist()
    for c, (g, k) in enumerate(zip(m["mols"].tolist(), m["confs"].tolist())):
        block = m["values"][starts[c]:starts[c + 1]]
        assert block.shape == (n_atoms[g], 3)
        assert torch.equal(block[:, 0], torch.full((n_atoms[g],), float(g), dtype=torch.float64))
        assert torch.equal(block[:, 1], torch.full((n_atoms[g],), float(k), dtype=torch.float64))
        assert torch.equal(block[:, 2], torch.arange(n_atoms[g], dtype=torch.float64))
    # determinism and edge cases of the assignment
    assert shard_molecules_by_cost(n_atoms, 1, 0).tolist() == list(range(len(n_atoms)))
    assert shard_molecules_by_cost([], 4, 2).tolist() == []
    parts = [shard_molecules_by_cost(np.full(10, 20), 4, r) for r in range(4)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(10)) and max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _butina_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import json

    import oracle
    from nvmolkit_amd.distributed import fused_butina_sharded, shard_bounds
    from tests import util

    x = util.clustered_fingerprints(257, 8, 9, seed=4)        # replicated on every rank

    def pairs_fn(xx, cutoff, shard, n_shards):                # CPU stand-in for nvmk_butina_pairs: a row band of the pass
        lo, hi = shard_bounds(len(x), n_shards, shard)
        c, p = oracle.neighbor_pairs(x, cutoff, lo, hi)
        return torch.from_numpy(c), torch.from_numpy(p)

    def rounds_fn(n, counts, pairs):                          # CPU stand-in for nvmk_butina_from_pairs
        return oracle.butina_from_pairs(n, counts.numpy(), pairs.numpy())

    got = fused_butina_sharded(torch.from_numpy(x.view(np.int32)), 0.35, return_centroids=True, pairs_fn=pairs_fn,
                               rounds_fn=rounds_fn)
    with open(os.path.join(out_dir, f"butina{rank}.json"), "w") as f:
        json.dump([list(map(list, got[0])), got[1], got[2]], f)
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_fused_butina_equals_single_process(tmp_path):
    """the all-pairs pass sharded over two ranks, degrees all-reduced, pair lists all-gathered, the
    round loop replicated — both ranks end with the single-process clustering."""
    import json

    world = 2
    mp.spawn(_butina_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, str(ROOT))
    import oracle
    from tests import util

    x = util.clustered_fingerprints(257, 8, 9, seed=4)
    want = oracle.butina_fused(x, 0.35)
    for r in range(world):
        clusters, cum, cent = json.load(open(tmp_path / f"butina{r}.json"))
        assert [tuple(c) for c in clusters] == want[0] and cum == want[1] and cent == want[2]
