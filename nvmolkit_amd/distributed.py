"""Multi-GPU sharding of the similarity path: one process per GPU, ``torch.distributed`` (nccl = RCCL on ROCm).

BASELINE.json configs[4]: the QUERY set is sharded (each rank owns its rows of the result), the REFERENCE set starts
sharded and is assembled with a single all-gather over xGMI — 32 MB per rank / 256 MB total at 1M x 2048 bit, ~2 ms
on a ring, invisible next to >1 s of similarity work per rank.  The reference has no collectives at
all; this is new capability.  Molecule batches (conformers, fingerprints) shard with no collective.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous, near-equal split of n rows; the first n % world_size ranks get one extra row."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def padded_shard_rows(n: int, world_size: int) -> int:
    """Rows every rank contributes to the all-gather (shards are padded to the largest one)."""
    return (n + world_size - 1) // world_size


def all_gather_rows(shard: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Assemble the (n_total, W) matrix from per-rank row shards laid out by :func:`shard_bounds`.

    One ``all_gather_into_tensor`` of equal (padded) shards, then the padding rows are dropped."""
    world = dist.get_world_size(group)
    rows = padded_shard_rows(n_total, world)
    buf = torch.zeros((rows, shard.shape[1]), dtype=shard.dtype, device=shard.device)
    buf[: shard.shape[0]] = shard
    gathered = torch.empty((rows * world, shard.shape[1]), dtype=shard.dtype, device=shard.device)
    dist.all_gather_into_tensor(gathered, buf, group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, world, r)
        parts.append(gathered[r * rows: r * rows + (hi - lo)])
    return torch.cat(parts, dim=0)


# ---- molecule batches (conformer generation / optimisation): shard, no data-path collective ----------------------

def shard_molecules_by_cost(n_atoms, world_size: int, rank: int, exponent: float = 2.0):
    """Indices of the molecules rank ``rank`` owns.  Molecules are independent, so the only multi-GPU question is
    balance: cost per BFGS iteration grows like atoms^2 (inverse-Hessian traffic), so molecules are dealt out largest
    first to the least loaded rank (LPT), deterministically — every rank computes the same assignment with no
    communication.  The reference round-robins batches over GPUs (src/etkdg.cpp:330-380, bfgs_mmff.cpp:139-157)."""
    import numpy as np

    return np.nonzero(molecule_owners_by_cost(n_atoms, world_size, exponent)[0] == rank)[0]


def molecule_owners_by_cost(n_atoms, world_size: int, exponent: float = 2.0):
    """(owner rank of every molecule, modelled load of every rank) of :func:`shard_molecules_by_cost`."""
    import numpy as np

    cost = np.asarray(n_atoms, dtype=np.float64) ** exponent
    order = np.lexsort((np.arange(len(cost)), -cost))  # largest first, index as tie-break
    load = np.zeros(world_size)
    owner = np.empty(len(cost), dtype=np.int64)
    for i in order:
        r = int(np.argmin(load))  # first minimum: deterministic
        owner[i] = r
        load[r] += cost[i]
    return owner, load


def merge_device_results(local, global_mol_ids, n_mols_total: int, group=None):
    """All-gather per-rank :class:`~nvmolkit_amd.types.Device3DResult` shards into one result ordered by (molecule,
    conformer) on every rank — the multi-GPU counterpart of the reference's ``finalizeOnTarget``
    (src/conformer/device_coord_collector.cpp) for ranks that are processes rather than threads.

    ``global_mol_ids[m]`` is the input-batch index of the shard's molecule m.  Not on the hot path: bookkeeping after
    the embarrassingly parallel work, a few padded all-gathers of the coordinates and per-conformer metadata."""
    from nvmolkit_amd.types import Device3DResult

    world = dist.get_world_size(group)
    values = local.values.torch()
    dev = values.device
    starts = local.atom_starts.torch().to(torch.int64)
    sizes = starts[1:] - starts[:-1]
    gmol = torch.as_tensor(global_mol_ids, dtype=torch.int64, device=dev)[local.mol_indices.torch().to(torch.int64)]
    conf = local.conf_indices.torch().to(torch.int64)
    has_e = local.energies is not None
    meta = torch.tensor([values.shape[0], sizes.numel()], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    max_atoms = max(int(m[0]) for m in metas)
    max_confs = max(int(m[1]) for m in metas)

    def gather(t: torch.Tensor, rows: int) -> list[torch.Tensor]:
        buf = torch.zeros((rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
        buf[: t.shape[0]] = t
        out = torch.empty((rows * world,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
        dist.all_gather_into_tensor(out, buf, group=group)
        return [out[r * rows:(r + 1) * rows] for r in range(world)]

    g_vals = gather(values, max_atoms)
    g_size, g_mol, g_conf = gather(sizes, max_confs), gather(gmol, max_confs), gather(conf, max_confs)
    g_e = gather(local.energies.torch(), max_confs) if has_e else None
    g_c = gather(local.converged.torch(), max_confs) if has_e else None
    vals = torch.cat([g_vals[r][: int(metas[r][0])] for r in range(world)])
    take = lambda g: torch.cat([g[r][: int(metas[r][1])] for r in range(world)])  # noqa: E731
    size, mol, cnf = take(g_size), take(g_mol), take(g_conf)
    order = torch.argsort(mol * (int(cnf.max()) + 1 if cnf.numel() else 1) + cnf, stable=True)
    old_starts = torch.cumsum(size, 0) - size
    size_o = size[order]
    new_starts = torch.zeros(size.numel() + 1, dtype=torch.int64, device=dev)
    new_starts[1:] = torch.cumsum(size_o, 0)
    rows = (torch.repeat_interleave(old_starts[order] - new_starts[:-1], size_o) +
            torch.arange(int(new_starts[-1]), device=dev))
    return Device3DResult(vals[rows], new_starts.to(torch.int32), mol[order].to(torch.int32), cnf[order].to(torch.int32),
                          local.gpu_id, n_mols_total,
                          energies=take(g_e)[order] if has_e else None, converged=take(g_c)[order] if has_e else None)


# ---- fused Butina with a row-sharded all-pairs pass ---------------------------------------------

def butina_pairs_gpu(x: torch.Tensor, cutoff: float, shard: int, n_shards: int, metric: str = "tanimoto", capacity: int | None = None):
    """This shard's band of the symmetric all-pairs pass on the GPU (``nvmk_butina_pairs``): (partial degrees int32[N],
    neighbour pairs int32[E, 2]) on ``x``'s device, original row numbers."""
    import ctypes

    from nvmolkit_amd import _native
    from nvmolkit_amd.clustering import _METRICS, _check_fingerprint_matrix

    _check_fingerprint_matrix("x", x)
    x = x.contiguous()
    n = x.shape[0]
    cap = int(capacity if capacity is not None else min(n * 128, (1 << 30) - 1))
    for attempt in range(2):
        counts = torch.zeros(max(n, 1), dtype=torch.int32, device=x.device)
        pairs = torch.empty((max(cap, 1), 2), dtype=torch.int32, device=x.device)
        n_pairs = ctypes.c_uint64(0)
        with torch.cuda.device(x.device):
            rc = _native.lib().nvmk_butina_pairs(_METRICS[metric], x.data_ptr(), n, x.shape[1] * 32, float(cutoff), int(shard),
                                                 int(n_shards), counts.data_ptr(), pairs.data_ptr(), cap, ctypes.byref(n_pairs),
                                                 _native.stream_ptr(None))
        # a graph denser than the default buffer (mean degree above ~256): the call reports how many pairs there are —
        # once more with exactly that capacity
        if rc == _native.ERR_OUT_OF_MEMORY and attempt == 0 and cap < n_pairs.value <= (1 << 31) - 1:
            del pairs
            cap = int(n_pairs.value)
            continue
        break
    _native.check(rc, "nvmk_butina_pairs")
    return counts[:n], pairs[: n_pairs.value]


def butina_from_pairs_gpu(n: int, counts: torch.Tensor, pairs: torch.Tensor):
    """The device-side round loop on an assembled neighbour graph (``nvmk_butina_from_pairs``) -> (clusters, cumulative
    sizes, centroids) like :func:`nvmolkit_amd.clustering.fused_butina` with ``return_centroids=True``."""
    import ctypes

    import numpy as np

    from nvmolkit_amd import _native

    counts = counts.to(torch.int32).contiguous()
    pairs = pairs.to(torch.int32).contiguous()
    idx = np.empty(max(n, 1), dtype=np.int32)
    offs = np.zeros(n + 1, dtype=np.int64)
    cent = np.empty(max(n, 1), dtype=np.int32)
    nc = ctypes.c_int64(0)
    with torch.cuda.device(counts.device):
        rc = _native.lib().nvmk_butina_from_pairs(n, counts.data_ptr(), pairs.data_ptr() if pairs.numel() else None,
                                                  pairs.shape[0], idx.ctypes.data, offs.ctypes.data, cent.ctypes.data,
                                                  ctypes.byref(nc), _native.stream_ptr(None))
    _native.check(rc, "nvmk_butina_from_pairs")
    k = nc.value
    bounds = offs[:k + 1].tolist()
    flat = memoryview(idx)  # tuple() of a slice makes the Python ints directly (as clustering.fused_butina)
    return [tuple(flat[bounds[c]:bounds[c + 1]]) for c in range(k)], bounds, cent[:k].tolist()


def fused_butina_sharded(x: torch.Tensor, cutoff: float, group=None, metric: str = "tanimoto", return_centroids: bool = False,
                         pairs_fn=None, rounds_fn=None):
    """Fused Butina over the ranks of ``group``: the fingerprint matrix ``x`` is replicated (256 MB at 1M x 2048 bit), the
    O(N^2) all-pairs pass is sharded by bands of tile rows of equal area, and the neighbour GRAPH it produces is exchanged
    once — one all-reduce of the degree vector (4 N bytes) and one all-gather of the pair lists (8 bytes per neighbour
    pair: ~180 MB at 1M rows, cutoff 0.3).  Every rank then runs the same deterministic device-side round loop on the
    assembled graph, so there is NO per-round collective (a (count, index) all-reduce per round, the textbook scheme,
    would be ~20 000 latency-bound collectives at 1M rows).  Returns on every rank exactly what the single-GPU
    :func:`nvmolkit_amd.clustering.fused_butina` returns.

    ``pairs_fn(x, cutoff, shard, n_shards) -> (counts, pairs)`` and ``rounds_fn(n, counts, pairs)`` default to the GPU
    entry points; the CPU test of the exchange injects the oracle's."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    pairs_fn = pairs_fn or (lambda xx, c, s, k: butina_pairs_gpu(xx, c, s, k, metric))
    rounds_fn = rounds_fn or butina_from_pairs_gpu
    n = x.shape[0]
    counts, pairs = pairs_fn(x, cutoff, rank, world)
    counts = counts.to(torch.int32).contiguous()
    pairs = pairs.to(torch.int32).reshape(-1, 2).contiguous()
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    sizes = torch.zeros(world, dtype=torch.int64, device=counts.device)
    sizes[rank] = pairs.shape[0]
    dist.all_reduce(sizes, op=dist.ReduceOp.SUM, group=group)
    rows = int(sizes.max().item())
    buf = torch.zeros((max(rows, 1), 2), dtype=torch.int32, device=counts.device)
    buf[: pairs.shape[0]] = pairs
    gathered = torch.empty((max(rows, 1) * world, 2), dtype=torch.int32, device=counts.device)
    dist.all_gather_into_tensor(gathered, buf, group=group)
    every = torch.cat([gathered[r * max(rows, 1): r * max(rows, 1) + int(sizes[r].item())] for r in range(world)])
    clusters, cum, cent = rounds_fn(n, counts, every)
    return (clusters, cum, cent) if return_centroids else (clusters, cum)
