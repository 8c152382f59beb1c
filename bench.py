#!/usr/bin/env python3
"""Headline benchmark: 1M x 1M 2048-bit Tanimoto cross-similarity on MI355X.

One "step" = one full pass of the dense similarity hot path (`nvmk_cross_tanimoto_f64`, float64
output, BASELINE.json configs[1]) over N_query x N_ref fingerprints that are already resident in
HBM.  The 8 TB result cannot exist at once, so the query rows are walked in chunks
whose N_chunk x N_ref double block is written to a reused HBM buffer — exactly what
`crossTanimotoSimilarityMemoryConstrained` does minus the PCIe copy.  Every pair is computed and
every double is stored.

Multi-GPU (`--gpus N`, launched by torch.distributed.run): BASELINE.json configs[4] — the query set
is sharded (weak scaling: 1M queries per rank), the reference set starts sharded N ways and is
assembled with ONE RCCL all-gather per step inside the timed region; no other collective.

Prints one JSON line (rank 0).  `roofline` is for the dense kernel (HBM-write bound, 8 B/pair);
`cpu_baseline` times the CPU oracle (a port, OpenMP popcount) on a bounded sample of the same data.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
SEED = 20260926


def synth_fingerprints(n: int, words: int, device, seed: int, n_centres: int | None = None,
                       density_range: tuple[float, float] | None = None, row_range: tuple[int, int] | None = None) -> torch.Tensor:
    """BASELINE.md cfg2 generator on the GPU: planted clusters, ~2.3 % density, <= 12 bit flips per row.
    ``density_range=(lo, hi)`` gives every cluster centre its own bit density instead (popcounts spread like real
    Morgan fingerprints of small to large molecules; used by tools/bench_butina.py --spread).
    ``row_range=(lo, hi)``: only those rows of the n-row set, with every block of 2^18 rows drawn from its own seed — a rank of
    a sharded run makes its own rows without materialising the whole set (every rank gets the same centres and the same rows
    for the same indices; the stream differs from the one-generator stream of the unsharded call)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lo_r, hi_r = row_range if row_range is not None else (0, n)
    nbits = words * 32
    n_centres = n_centres or max(1, n // 50)
    if density_range is None:
        centres = torch.rand((n_centres, nbits), generator=g, device=device) < 0.023
    else:
        dens = density_range[0] + (density_range[1] - density_range[0]) * torch.rand((n_centres, 1), generator=g, device=device)
        centres = torch.rand((n_centres, nbits), generator=g, device=device) < dens
    weights = (torch.ones(32, dtype=torch.int32, device=device) << torch.arange(32, dtype=torch.int32, device=device))
    centres_packed = (centres.reshape(n_centres, words, 32).to(torch.int32) * weights).sum(dim=2, dtype=torch.int32)
    step = 1 << 18
    if row_range is not None:
        parts = []
        for lo in range((lo_r // step) * step, hi_r, step):
            hi = min(n, lo + step)
            gb = torch.Generator(device=device)
            gb.manual_seed(seed * 1_000_003 + lo // step + 1)
            block = _planted_rows(hi - lo, centres_packed, n_centres, nbits, gb, device)
            parts.append(block[max(lo_r - lo, 0): min(hi_r, hi) - lo])
        return torch.cat(parts) if parts else torch.empty((0, words), dtype=torch.int32, device=device)
    out = torch.empty((n, words), dtype=torch.int32, device=device)
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        out[lo:hi] = _planted_rows(hi - lo, centres_packed, n_centres, nbits, g, device)
    return out


def _planted_rows(m: int, centres_packed: torch.Tensor, n_centres: int, nbits: int, g, device) -> torch.Tensor:
    """m rows: a random centre each, then up to 12 bit flips."""
    owner = torch.randint(0, n_centres, (m,), generator=g, device=device)
    rows = centres_packed[owner].clone()
    k = torch.randint(0, 13, (m, 1), generator=g, device=device)
    pos = torch.randint(0, nbits, (m, 12), generator=g, device=device)
    active = torch.arange(12, device=device).unsqueeze(0) < k
    word = pos // 32
    bit = (torch.ones_like(pos, dtype=torch.int32) << (pos % 32).to(torch.int32)) * active.to(torch.int32)
    for j in range(12):  # XOR flips one at a time (duplicates cancel, like repeated flips)
        rows.scatter_(1, word[:, j:j + 1], rows.gather(1, word[:, j:j + 1]) ^ bit[:, j:j + 1])
    return rows


def cpu_baseline(ref_words: np.ndarray, target_seconds: float) -> dict:
    """Time the CPU port (oracle/oracle_similarity.c: OpenMP over reference blocks, AVX-512 VPOPCNTDQ popcounts 2 x 8 pairs at a
    time where the host has them, IEEE double division) on row blocks of the same workload: all host threads for about
    ``target_seconds``, then one thread for a fifth of that.  Blocks of 2048 query rows keep the parallel part of a call
    (10^8 pairs and more) well above its serial set-up."""
    import oracle

    threads = oracle.num_threads()
    lib = oracle.lib()
    ref_words = ref_words[: min(len(ref_words), 100_000)]
    m = ref_words.shape[0]
    block = min(max(2048, 16 * threads), m)
    out = np.empty((block, m), dtype=np.float64)  # first touched by the (untimed) first call, by the threads that keep writing it

    def run(n_threads: int, seconds: float):
        lib.orc_cross_similarity_f64(0, np.ascontiguousarray(ref_words[:block]), block, ref_words, m, ref_words.shape[1], out, m, n_threads)
        pairs, row, t0 = 0, 0, time.perf_counter()
        while True:
            if row + block > m:
                row = 0
            a = np.ascontiguousarray(ref_words[row:row + block])
            lib.orc_cross_similarity_f64(0, a, block, ref_words, m, ref_words.shape[1], out, m, n_threads)
            pairs += block * m
            row += block
            dt = time.perf_counter() - t0
            if dt >= seconds:
                return pairs, dt

    pairs, dt = run(0, target_seconds)
    pairs1, dt1 = run(1, max(1.0, target_seconds / 5.0))
    return {
        "value": pairs / dt,
        "unit": "pairs/s",
        "cores": threads,
        "kind": "port",
        "one_thread_value": pairs1 / dt1,
        "vector_popcount": bool(lib.orc_have_vpopcnt()),
        "sample": f"{pairs // m} query rows x {m} reference rows of the same synthetic set ({pairs:.3g} pairs, "
                  f"{dt:.1f} s, oracle/oracle_similarity.c with OpenMP on {threads} threads; then {pairs1:.3g} pairs in "
                  f"{dt1:.1f} s on one thread); float64 matrix written to host memory like the GPU writes it to HBM",
    }


def kernel_source_digest(names=("similarity_mfma.hip", "fp4.h", "similarity.hip")) -> str:
    """sha256 over the dense-similarity kernel sources: a PMC traffic file is only quoted for the code it was measured on."""
    import hashlib

    h = hashlib.sha256()
    for name in names:
        h.update((ROOT / "nvmolkit_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()


def conformer_source_digest() -> str:
    """sha256 over the conformer kernels' sources (as tools/profile_conformer_traffic.sh computes it)."""
    return kernel_source_digest(("minimize.hip", "minimize_team.hip", "bfgs_common.h", "bfgs_device.inc", "hess_pass.h", "ff_terms.h", "ff_grad.h",
                                 "etkdg.hip", "table_build.cpp"))


def butina_block(n: int, words: int, device, cpu_seconds: float) -> dict:
    """Fused Butina (cutoff 0.3 = similarity threshold 0.7, BASELINE.json configs[1]) on n planted-cluster fingerprints,
    with the roofline of its dominant kernel (the FP4 matrix-core neighbour-count pass) and the C oracle timed on a sample."""
    from nvmolkit_amd.clustering import fused_butina

    xb = synth_fingerprints(n, words, device, SEED)
    fused_butina(xb, 0.3)  # warm-up at full size: scratch pools, hipcub temp storage
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tb = time.perf_counter()
    ev0.record()
    clusters, sizes = fused_butina(xb, 0.3)
    ev1.record()
    torch.cuda.synchronize()
    tb = time.perf_counter() - tb
    pairs = n * (n + 1) / 2.0                       # the symmetric all-pairs pass evaluates every unordered pair once
    flops = pairs * 2.0 * words * 32                # one multiply-add per bit of the fingerprint
    out = {"n": n, "cutoff": 0.3, "seconds": tb, "n_clusters": len(clusters), "fingerprints_per_s": n / tb,
           "timing": "second call on the same set (whole call: all-pairs pass + CSR + device round loop + host lists)",
           "roofline": {"bound": "mfma", "achieved": flops / tb / 1e12, "peak": 10000.0, "unit": "TFLOP/s",
                        "frac": flops / tb / 1e12 / 10000.0, "traffic": None,
                        "kernel": "nvmk::fp4::neighbor_count_panel_kernel at this size (row panels, csrc/count_panel.inc; below 262 144 rows neighbor_count_mfma_kernel) — FP4 e2m1 x e2m1 -> f32, exact 0/1 products",
                        "note": "algorithmic flops = n (n + 1) / 2 pairs x 2 x fp_bits, divided by the WHOLE call's wall time "
                                "(conservative: the pass is ~83 % of the call, 0.427 of 0.517 s; the rest is the CSR build, the rounds and the Python "
                                "lists the API returns); peak = ~10 PF dense FP4 MFMA "
                                "(MI355X_MICROARCH.md; 9.1 PF measured there)"}}
    out["pairs_per_s"] = pairs / tb
    if cpu_seconds > 0:
        import oracle

        # The CPU's cost is the same O(n^2) neighbour pass (the round loop is noise next to it): time the oracle's OpenMP
        # popcount pass on a slab of the same set and quote it in pairs/s, the unit both sides share.
        m = min(n, 100_000)
        sub = xb[:m].cpu().numpy().view(np.uint32)
        rows = sub[: max(256, min(m, int(cpu_seconds * 2.0e8 / m)))]      # sized for roughly cpu_seconds of work
        t0 = time.perf_counter()
        oracle.neighbor_counts(rows, sub, np.float32(0.7))
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": len(rows) * m / dt, "unit": "pairs/s", "cores": oracle.num_threads(), "kind": "port",
                               "sample": f"{len(rows)} x {m} rows of the same set through oracle/oracle_similarity.c "
                                         f"orc_neighbor_counts (the all-pairs neighbour pass, OpenMP popcount), {dt:.1f} s; the "
                                         f"GPU figure to compare is pairs_per_s = n (n + 1) / 2 / seconds"}
    return out


def cfg1_block(device, cpu_seconds: float) -> dict:
    """BASELINE.json configs[0]: the reference's 10 000 benchmark SMILES (benchmarks/data/chembl_10k.smi, kept as input
    data under tests/golden/) -> Morgan r = 2 / 2048 bit -> 10k x 10k Tanimoto, end to end without RDKit: the library's own
    SMILES ingestion on the host, the Morgan kernels and the dense similarity kernel on the GPU."""
    from nvmolkit_amd.fingerprints import MorganFingerprintGenerator, SmilesSet
    from nvmolkit_amd.similarity import crossTanimotoSimilarity

    path = ROOT / "tests" / "golden" / "chembl_10k.smi"
    smiles = [line.split()[0] for line in path.read_text().splitlines() if line.strip()]
    gen = MorganFingerprintGenerator(radius=2, fpSize=2048)
    gen.GetFingerprintsFromSmiles(smiles[:512]).torch()                # warm-up: library, staging pools
    crossTanimotoSimilarity(gen.GetFingerprintsFromSmiles(smiles[:512]).torch()).torch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mols = SmilesSet(smiles)
    t_parse = time.perf_counter() - t0
    fps = gen.GetFingerprintsFromSmiles(mols).torch()
    torch.cuda.synchronize()
    t_fp = time.perf_counter() - t0
    sim = crossTanimotoSimilarity(fps).torch()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    n = len(smiles)
    out = {"molecules": n, "ingested": int((mols.status == 0).sum()), "seconds": t_all, "smiles_parse_seconds": t_parse,
           "smiles_to_fingerprints_seconds": t_fp, "similarity_seconds": t_all - t_fp,
           "fingerprints_per_s": n / t_fp, "similarity_pairs_per_s": float(n) * n / (t_all - t_fp),
           "data": "tests/golden/chembl_10k.smi (the reference's benchmarks/data/chembl_10k.smi)",
           "note": "one cold pass: SMILES strings on the host -> (10000, 10000) float64 Tanimoto matrix on the GPU"}
    if cpu_seconds > 0:
        import oracle

        size = np.maximum(mols.n_atoms, mols.n_bonds)
        t0 = time.perf_counter()
        want = np.zeros((n, 64), dtype=np.uint32)
        lo = 0
        for stride in (32, 64, 128, 256, 512, 1024):
            idx = np.flatnonzero((size >= lo) & (size < stride) & (mols.status == 0))
            lo = stride
            if len(idx):
                want[idx] = oracle.morgan_fingerprints(*mols.morgan_inputs(idx, stride), stride, 2, 2048)
        t_cpu_fp = time.perf_counter() - t0
        ref = oracle.cross_similarity(want, want)
        t_cpu = time.perf_counter() - t0
        rows = np.r_[0:32, n - 32:n]
        out["cpu_baseline"] = {"value": t_cpu, "unit": "s", "cores": oracle.num_threads(), "kind": "port",
                               "fingerprint_seconds": t_cpu_fp, "similarity_seconds": t_cpu - t_cpu_fp,
                               "sample": "the whole configuration: oracle/oracle_morgan.c on the same graphs + oracle/oracle_similarity.c "
                                         "(OpenMP) for the 10k x 10k matrix"}
        out["matches_cpu_port"] = bool(np.array_equal(fps.cpu().numpy().view(np.uint32), want)
                                       and np.array_equal(sim[rows].cpu().numpy(), ref[rows]))
    return out


def morgan_block(device) -> dict:
    """the Morgan kernel (radius 2, 2048 bit) on RESIDENT inputs, per size bucket — the molecules of the
    reference's benchmark file that fall into the 32 / 64 / 128-slot buckets (atoms and bonds below the slot count: the
    reference's bucketing rule), their invariant arrays made by the library's ingestion and tiled on the device to 2^20 (32, 64)
    or 2^18 (128) molecules; three launches per bucket timed with events on the launch stream."""
    from nvmolkit_amd import _native
    from nvmolkit_amd.fingerprints import SmilesSet

    lib = _native.lib()
    mols = SmilesSet.from_file(str(ROOT / "tests" / "golden" / "chembl_10k.smi"))
    size = np.maximum(mols.n_atoms, mols.n_bonds)
    stream = torch.cuda.current_stream()
    out, lo = {"radius": 2, "fp_bits": 2048, "data": "tests/golden/chembl_10k.smi by bucket, tiled on the device", "buckets": {}}, 0
    for stride, target in ((32, 1 << 20), (64, 1 << 20), (128, 1 << 18)):
        idx = np.flatnonzero((size >= lo) & (size < stride) & (mols.status == 0))
        lo = stride
        if len(idx) == 0:
            continue
        host = mols.morgan_inputs(idx, stride)
        reps = (target + len(idx) - 1) // len(idx)
        d = [torch.from_numpy(np.ascontiguousarray(a.view(np.int32) if a.dtype == np.uint32 else a)).to(device) for a in host]
        d = [t.repeat((reps,) + (1,) * (t.dim() - 1))[:target].contiguous() for t in d]
        fps = torch.empty((target, 64), dtype=torch.int32, device=device)

        def launch():
            _native.check(lib.nvmk_morgan_from_invariants(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                                                          None, target, stride, 2, 2048, fps.data_ptr(), int(stream.cuda_stream)))

        launch()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for _ in range(3):
            launch()
        ev1.record(stream)
        torch.cuda.synchronize()
        dt = ev0.elapsed_time(ev1) * 1e-3 / 3
        nbytes = sum(t.numel() * t.element_size() for t in d) + fps.numel() * 4
        out["buckets"][str(stride)] = {"molecules": target, "distinct_molecules": int(len(idx)), "mean_atoms": float(mols.n_atoms[idx].mean()),
                                       "seconds_per_launch": dt, "mols_per_s": target / dt,
                                       "algorithmic_GB_per_s": nbytes / dt / 1e9,   # five input arrays read once + fingerprints written once
                                       "frac_of_hbm_peak": nbytes / dt / 1e9 / HBM_PEAK_GBPS}
        del d, fps
    out["note"] = ("kernel nvmk::morgan::morgan_kernel: one wave per molecule (two molecules per wave in the 32 bucket), all state in "
                   "LDS; bound by LDS traffic and latency, not by HBM (DESIGN.md 4.3) - the HBM fraction is given for scale")
    return out


BFGS_KIND_NAMES = {0: "dg", 1: "etk", 2: "mmff"}


def conformer_library(n_mols: int, world: int, rank: int, shared: bool = False):
    """The synthetic drug-like molecules (generated in forked worker processes) and the seconds it took.
    ``shared`` (multi-rank runs): only rank 0 generates — the start-up of --gpus 8 costs one generation, not eight on an eighth
    of the cores each — and the others get ``None`` here and the set itself from :func:`share_library` once the process group
    is up; every rank then works on the same molecules (weak scaling: with its own seed; strong scaling: on its share of them)."""
    from nvmolkit_amd import synthetic

    t0 = time.perf_counter()
    if shared and rank != 0:
        return None, 0.0
    procs = max(1, (os.cpu_count() or 2) // 2)
    return synthetic.druglike_library(n_mols, seed=SEED, processes=min(procs, 64)), time.perf_counter() - t0


def share_library(library, rank: int):
    """Rank 0's molecule library to every rank, through a file in a PRIVATE directory (``mkdtemp``: mode 0700, fresh name) whose
    path travels over the process group — not a predictable name in world-writable /dev/shm that a stale run or another user
    could have put there first.  The others check that the directory is theirs alone before unpickling, and rank 0 removes it
    once everybody has read it.  Call after ``init_process_group``."""
    import pickle
    import shutil
    import tempfile

    import torch.distributed as dist

    box = [None]
    if rank == 0:
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        box[0] = tempfile.mkdtemp(prefix="nvmk_bench_", dir=base)
        with open(os.path.join(box[0], "library.pkl"), "wb") as f:
            pickle.dump(library, f, protocol=pickle.HIGHEST_PROTOCOL)
    dist.broadcast_object_list(box, src=0)
    try:
        if rank != 0:
            st = os.stat(box[0])
            if st.st_uid != os.getuid() or (st.st_mode & 0o077):
                raise SystemExit(f"rank {rank}: {box[0]} is not a private directory of this user; refusing to load from it")
            with open(os.path.join(box[0], "library.pkl"), "rb") as f:
                library = pickle.load(f)
        dist.barrier()
    finally:
        if rank == 0:
            shutil.rmtree(box[0], ignore_errors=True)
    return library


def strong_scaling_share(library, total: int, world: int, rank: int):
    """configs[3]: ONE job of ``total`` molecules dealt over the ranks by cost (distributed.shard_molecules_by_cost: largest first
    to the least loaded rank, the same assignment computed on every rank, no communication).  The job is the generated set
    repeated as often as needed (every instance is embedded on its own: own attempts, own random start coordinates).
    Returns (this rank's molecules, modelled cost of every rank's share)."""
    from nvmolkit_amd.distributed import molecule_owners_by_cost

    sizes = np.array([m["embed"]["n_atoms"] for m in library])
    owner, load = molecule_owners_by_cost(sizes[np.arange(total) % len(library)], world)
    return [library[int(i) % len(library)] for i in np.nonzero(owner == rank)[0]], [float(x) for x in load]


def conformer_block(n_mols: int, confs: int, mmff_iters: int, device, world: int, rank: int, cpu_seconds: float,
                    library=None, t_library: float = 0.0, collectives: bool = False, strong_total: int = 0,
                    data: str | None = None, resident_rerun: bool = True, cpu_sample_max_atoms: int | None = None,
                    pmc_file: str | None = "pmc_hbm_traffic_conformers.json", cpu_shared: dict | None = None) -> dict:
    """BASELINE.json configs[2] (and [3] when world > 1: every rank embeds and optimises its own n_mols molecules, no
    data-path collective): ETKDG (`confs` conformers per molecule) DEVICE-chained into MMFF94 on the synthetic drug-like
    set of nvmolkit_amd/synthetic.py.  Roofline = the fused BFGS kernels' inverse-Hessian traffic."""
    from nvmolkit_amd import _native, mmffOptimization
    from nvmolkit_amd.embedMolecules import FlatMolecule, FlatMoleculeSet, embed_flat
    from nvmolkit_amd.types import CoordinateOutput

    if library is None:
        library, t_library = conformer_library(n_mols, world, rank)
    share_cost = None
    if strong_total > 0:
        library, share_cost = strong_scaling_share(library, strong_total, world, rank)
        n_mols = len(library)
    lib = _native.lib()
    # warm-up (untimed, like the headline's warm-up steps): the pipeline once at the timed run's batch size, with another seed —
    # module load, the table builder's pinned staging ring, and the stream-ordered pool then holds blocks of the sizes the timed
    # run asks for (per-workgroup inverse-Hessian slots of every class, batch-sized work arrays).  A 1000-molecule warm-up (the
    # round-5 closing line) left the pool to grow during the first timed batch: ETKDG 1.99 s in the timed run against 1.91 s in
    # the run after it.  Everything the timed region builds — molecule set, term tables — is built again from the host arrays.
    # (molecules beyond 128 atoms — only the opt-in whole-file block has them — stay out of the warm-up: they take minutes)
    warm_lib = [m for m in library[:min(10000, n_mols)] if m["embed"]["n_atoms"] <= 128] or library[:1]
    warm = embed_flat(FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in warm_lib], device=device), confs_per_molecule=confs,
                      max_iterations=10, seed=99, output=CoordinateOutput.DEVICE)
    mmffOptimization.optimize_device(mmffOptimization.resident_tables([m["mmff"] for m in warm_lib], device, wait=False), warm,
                                     max_iters=mmff_iters)
    del warm
    # (round 6: with the history form the large molecules take seconds, not minutes — they get a warm-up of their own, two
    # conformers each, so that the team classes' record slots are in the pool before the clock starts)
    big_lib = [m for m in library[:min(10000, n_mols)] if m["embed"]["n_atoms"] > 128]
    if big_lib:
        warm = embed_flat(FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in big_lib], device=device), confs_per_molecule=2,
                          max_iterations=10, seed=99, output=CoordinateOutput.DEVICE)
        mmffOptimization.optimize_device(mmffOptimization.resident_tables([m["mmff"] for m in big_lib], device, wait=False), warm,
                                         max_iters=mmff_iters)
        del warm
    stats = torch.zeros(64, dtype=torch.int64, device=device)
    _native.check(lib.nvmk_bfgs_set_stats(stats.data_ptr()))
    if collectives:
        import torch.distributed as dist

        dist.barrier()
    torch.cuda.synchronize()
    # (the interpreter's cyclic collector stays out of the timed regions, as in timeit: the region allocates ~10^5 small objects
    # next to a heap of millions, and a full collection that happens to start inside it costs 0.1-0.15 s — it landed in the
    # ChEMBL block of one closing run of round 5 and in the synthetic block of the next)
    import gc

    gc.collect()
    gc.disable()
    # The timed region starts with the molecules as the generator left them (per-molecule numpy term arrays: what an RDKit
    # molecule is to the reference) and holds everything the reference's benchmark times inside EmbedMolecules +
    # MMFFOptimizeMoleculesConfs (benchmarks/etkdg_bench.py:108-124): table assembly on the host threads, uploads, both GPU stages.
    # The MMFF tables are assembled on a host thread and stream of their own while ETKDG runs (resident_tables(wait=False)).
    t0 = time.perf_counter()
    molset = FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in library], device=device)
    t_molset = time.perf_counter() - t0   # host side of the assembly; its last uploads are still in flight
    # (after=molset: the MMFF assembly — seconds of every host thread on the file with its peptides — starts when the molecule
    # set's own asynchronous fill is through, which the first ETKDG batch is waiting for; started together, the whole file's
    # ETKDG took 21.1 s against 18.2 s, tools/experiments/e2e_overlap_probe.py)
    pending = mmffOptimization.resident_tables([m["mmff"] for m in library], device, wait=False, after=molset)
    dev = embed_flat(molset, confs_per_molecule=confs, max_iterations=10, seed=1 + rank, output=CoordinateOutput.DEVICE)
    torch.cuda.synchronize()
    t_embed = time.perf_counter() - t0
    t1 = time.perf_counter()
    tables = pending.result()
    t_tables_wait = time.perf_counter() - t1  # what of the MMFF assembly ETKDG did not hide
    opt = mmffOptimization.optimize_device(tables, dev, max_iters=mmff_iters)
    torch.cuda.synchronize()
    t_mmff = time.perf_counter() - t1
    wall = time.perf_counter() - t0
    _native.check(lib.nvmk_bfgs_set_stats(None))
    # the same job again on the tables that are now resident (rounds 2-4 reported only this figure): same seed, same work
    # (not for the whole benchmark file, whose single run takes most of a minute)
    wall_resident = r_embed = float("nan")
    same_bits = None
    if resident_rerun:
        torch.cuda.synchronize()
        r0 = time.perf_counter()
        dev_r = embed_flat(molset, confs_per_molecule=confs, max_iterations=10, seed=1 + rank, output=CoordinateOutput.DEVICE)
        torch.cuda.synchronize()
        r_embed = time.perf_counter() - r0
        opt_r = mmffOptimization.optimize_device(tables, dev_r, max_iters=mmff_iters)
        torch.cuda.synchronize()
        wall_resident = time.perf_counter() - r0
        same_bits = bool(torch.equal(opt_r.values.torch(), opt.values.torch()))
        del dev_r, opt_r
    gc.enable()
    t_flatten = t_molset + t_tables_wait
    n_conf = dev.num_conformers
    converged = int(opt.converged.torch().sum().item())
    if collectives:
        own_wall = wall
        t = torch.tensor([wall, t_embed, t_mmff, wall_resident, r_embed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        c = torch.tensor([n_conf, converged, n_mols], dtype=torch.int64, device=device)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        walls = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
        dist.all_gather(walls, torch.tensor([own_wall], dtype=torch.float64, device=device))
        per_rank_wall = [float(w.item()) for w in walls]
        wall, t_embed, t_mmff, wall_resident, r_embed = (float(x) for x in t.tolist())
        n_conf, converged, total_mols = (int(x) for x in c.tolist())
    else:
        total_mols = n_mols
        per_rank_wall = [wall]
    st = stats.cpu().numpy().reshape(8, 8)
    per_kind = {BFGS_KIND_NAMES[k]: {"systems": int(st[k, 0]), "iterations": int(st[k, 1]), "algorithmic_bytes": int(st[k, 2]),
                                     "energy_evaluations": int(st[k, 3]), "hbm_resident_bytes": int(st[k, 4]),
                                     # (round 6: minimisations whose inverse Hessian was kept as the history of its updates count the
                                     # bytes of the pairs they read and wrote above; this is what the packed triangle would have cost)
                                     "packed_triangle_bytes_of_the_same_iterations": int(st[k, 5]), "minimisations_in_history_form": int(st[k, 6]),
                                     "mean_iterations": float(st[k, 1]) / max(int(st[k, 0]), 1)} for k in BFGS_KIND_NAMES}
    algo = float(sum(v["algorithmic_bytes"] for v in per_kind.values()))
    requested = float(sum(v["hbm_resident_bytes"] for v in per_kind.values()))
    # HBM bytes of the block from the PMC passes committed under profiles/ (tools/profile_conformer_traffic.sh: separate
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs on 2000 molecules of the same set, FETCH_SIZE doubled as the guide's
    # gfx950 note prescribes), scaled by the conformers of this run; only while the kernel sources are the ones measured
    traffic, traffic_src = None, None
    # (pmc_file: the counter file of THIS block's molecule set — the synthetic set's ratios say nothing about the ChEMBL file's teams)
    for pmc in sorted((ROOT / "profiles").glob(f"r*/{pmc_file}"), reverse=True) if pmc_file else ():
        c = json.loads(pmc.read_text())
        if c.get("kernel_source_sha256") == conformer_source_digest() and c.get("by_kind"):
            # per kind: (bytes read + bytes written past the L2s) / (bytes the passes requested from HBM) of the PMC run, whose
            # requested bytes come from the kernels' counters of that SAME run — applied to this run's requested bytes (a
            # 2000-molecule run does more iterations per conformer than a 10 000-molecule one: bytes per conformer do not carry over)
            traffic = sum(0.5 * per_kind[k]["hbm_resident_bytes"] * (c["by_kind"][k]["read_ratio"] + c["by_kind"][k]["write_ratio"])
                          for k in per_kind if k in c["by_kind"])
            traffic_src = (f"{pmc.relative_to(ROOT)}: per kind (2 x FETCH_SIZE + WRITE_SIZE) / requested bytes of a {c.get('molecules')}-molecule "
                           f"run of the same set (reads " + " / ".join(f"{c['by_kind'][k]['read_ratio']:.2f}" for k in ("dg", "etk", "mmff")) +
                           ", writes " + " / ".join(f"{c['by_kind'][k]['write_ratio']:.2f}" for k in ("dg", "etk", "mmff")) +
                           " x requested for DG / ETK / MMFF) x this run's requested bytes, kernel source hash matches")
            break
    out = {"metric": f"mols/s ETKDG({confs} confs) + MMFF94 optimise (maxIters {mmff_iters})", "value": total_mols / wall,
           "unit": "mols/s", "n_gpus": world, "molecules": total_mols, "confs_per_molecule": confs,
           "mean_atoms": float(np.mean([m["embed"]["n_atoms"] for m in library])), "conformers": n_conf,
           "etkdg_seconds": t_embed, "etkdg_conformers_per_s": n_conf / t_embed, "mmff_seconds": t_mmff,
           "mmff_conformers_per_s": n_conf / t_mmff, "mmff_converged_fraction": converged / max(n_conf, 1),
           "timed_region": "per-molecule host term arrays -> table assembly (library's host threads, pinned staging, chunked uploads; "
                           "the MMFF tables on a thread and stream of their own under ETKDG) -> ETKDG -> MMFF, end to end",
           "library_generation_seconds": t_library,            # outside the clock: the generator stands in for reading molecules
           "table_assembly_host_seconds": t_molset,            # inside the clock (part of etkdg_seconds): FlatMoleculeSet's host side
           "mmff_tables_wait_seconds": t_tables_wait,          # inside the clock (part of mmff_seconds): MMFF assembly not hidden by ETKDG
           "flatten_and_table_upload_seconds": t_flatten,      # the two above: what of the flattening is on the critical path
           "resident_tables_value": (total_mols / wall_resident) if resident_rerun else None,  # the same job with both table sets already resident (rounds 2-4's figure)
           "resident_tables_seconds": wall_resident if resident_rerun else None, "resident_tables_etkdg_seconds": r_embed if resident_rerun else None,
           "resident_tables_run_gave_the_same_bits": same_bits,
           "scaling": "strong" if strong_total > 0 else "weak", "per_rank_seconds": per_rank_wall,
           "imbalance_max_over_mean": max(per_rank_wall) / (sum(per_rank_wall) / len(per_rank_wall)),
           "modelled_cost_share_max_over_mean": (max(share_cost) / (sum(share_cost) / len(share_cost))) if share_cost else None,
           "bfgs": per_kind,
           "data": data or ("synthetic drug-like molecules (rings + chains + hydrogens, bounds / ETK / MMFF tables derived from one "
                            "generated 3-D geometry; nvmolkit_amd/synthetic.py) — real SMILES need RDKit"),
           # achieved / frac: the bytes that crossed the L2s (PMC ratios of the same kernel sources applied to this run's requested
           # bytes) when such a file exists, else the bytes the inverse-Hessian passes requested from HBM — never the algorithmic
           # figure, which also counts the rows served from LDS
           "roofline": {"bound": "hbm", "achieved": (traffic if traffic else requested) / wall / 1e9 / world, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": (traffic if traffic else requested) / wall / 1e9 / HBM_PEAK_GBPS / world,
                        "frac_is": "frac_measured_traffic" if traffic else "frac_hbm_requested",
                        "traffic": traffic, "traffic_source": traffic_src,
                        # which bytes a fraction counts: every byte of the packed inverse Hessians the iterations stand for
                        # (rows served from LDS included) / only those the passes request from HBM / what crossed the L2s
                        "frac_algorithmic": algo / wall / 1e9 / HBM_PEAK_GBPS / world,
                        "frac_hbm_requested": requested / wall / 1e9 / HBM_PEAK_GBPS / world,
                        "frac_measured_traffic": (traffic / wall / 1e9 / HBM_PEAK_GBPS / world) if traffic else None,
                        "traffic_over_requested": (traffic / requested) if traffic and requested else None,
                        "traffic_over_algorithmic": (traffic / algo) if traffic and algo else None,
                        "hbm_bytes_requested_by_the_hessian_pass": requested,
                        "kernel": "nvmk::minim::bfgs_kernel<DG|ETK|MMFF> and, for systems of 656 coordinates or more, bfgs_team_kernel<DG|ETK|MMFF> "
                                  "(together > 99 % of the GPU time of this block)",
                        "packed_triangle_bytes_of_the_same_iterations": float(sum(v["packed_triangle_bytes_of_the_same_iterations"] for v in per_kind.values())),
                        "note": "algorithmic bytes = what the inverse-Hessian work of the block has to move, counted by the kernels "
                                "themselves (nvmk_bfgs_set_stats): per BFGS iteration 8 n (n + 2) (read + write of the packed "
                                "triangle) for a system that keeps the triangle, 16 n x (pairs stored so far) for a team system "
                                "that keeps the history of its rank-2 updates instead (each pair read once per iteration, written "
                                "once); divided by the block's wall time.  Rows of a triangle that stay in LDS never reach HBM, so "
                                "the bytes requested from HBM are the smaller figure beside it.  The history form moves FEWER bytes for "
                                "the same iterations (packed_triangle_bytes_of_the_same_iterations is what the triangle would have "
                                "cost), so this fraction went DOWN in round 6 while mols/s went up: it prices the memory system, "
                                "not the algorithm"}}
    if rank == 0 and world == 1 and cpu_seconds > 0 and cpu_shared is not None and "cpu_baseline" in cpu_shared:
        out["cpu_baseline"] = cpu_shared["cpu_baseline"]  # (the two ChEMBL blocks share ONE sample: the same molecules, timed once)
    elif rank == 0 and world == 1 and cpu_seconds > 0:
        from nvmolkit_amd.forcefield import MMFF as KIND_MMFF, stack_molecule_tables
        from oracle import ffc
        import oracle

        threads = oracle.num_threads()
        m = int(min(n_mols, max(16, 2 * min(threads, 128))))  # (bounded: ~10-30 s of CPU work whatever the host's thread count)
        # (the ChEMBL blocks: the sample is taken from the molecules of at most cpu_sample_max_atoms atoms — one 500-atom molecule
        # alone keeps a CPU thread busy for minutes)
        pool = library if cpu_sample_max_atoms is None else [x for x in library if x["embed"]["n_atoms"] <= cpu_sample_max_atoms]
        m = min(m, len(pool))
        sub = pool[:m]
        mols = [FlatMolecule(**x["embed"]) for x in sub]
        c0 = time.perf_counter()
        coords, counts, slots, fails, iters = ffc.etkdg_embed(mols, confs_per_molecule=confs, max_iterations=10, seed=1)
        c_embed = time.perf_counter() - c0
        n_at = np.array([x["embed"]["n_atoms"] for x in sub])
        sys_mol = np.repeat(np.arange(m), counts).astype(np.int32)
        a_s = np.concatenate([[0], np.cumsum(n_at[sys_mol])])
        pos = np.concatenate([coords[slots[i]:slots[i] + 3 * n_at[i] * counts[i]] for i in range(m)])
        batch = ffc.Batch(KIND_MMFF, a_s, stack_molecule_tables(KIND_MMFF, [x["mmff"] for x in sub]), system_mol=sys_mol)
        c1 = time.perf_counter()
        _, _, st_c, it_c = batch.minimize(pos, max_iters=mmff_iters)
        c_mmff = time.perf_counter() - c1
        # the same port on ONE thread, on the first two molecules (all-thread and one-thread numbers)
        one = None
        try:
            oracle.set_num_threads(1)
            m1 = min(2, m)
            o0 = time.perf_counter()
            coords1, counts1, slots1, _, _ = ffc.etkdg_embed(mols[:m1], confs_per_molecule=confs, max_iterations=10, seed=1)
            sys1 = np.repeat(np.arange(m1), counts1).astype(np.int32)
            as1 = np.concatenate([[0], np.cumsum(n_at[sys1])])
            pos1 = np.concatenate([coords1[slots1[i]:slots1[i] + 3 * n_at[i] * counts1[i]] for i in range(m1)])
            ffc.Batch(KIND_MMFF, as1, stack_molecule_tables(KIND_MMFF, [x["mmff"] for x in sub[:m1]]), system_mol=sys1).minimize(pos1, max_iters=mmff_iters)
            one = {"value": m1 / (time.perf_counter() - o0), "molecules": m1}
        finally:
            oracle.set_num_threads(0)
        out["cpu_baseline"] = {"value": m / (c_embed + c_mmff), "unit": "mols/s", "cores": threads, "kind": "port",
                               "one_thread_value": one["value"] if one else None,
                               "one_thread_sample": f"first {one['molecules']} molecules x {confs} conformers" if one else None,
                               "etkdg_seconds": c_embed, "mmff_seconds": c_mmff, "conformers": int(counts.sum()),
                               "mmff_converged_fraction": float((st_c == 0).mean()),
                               "sample": f"first {m} molecules of the same set" + (f" with at most {cpu_sample_max_atoms} atoms" if cpu_sample_max_atoms else "") +
                                         f" x {confs} conformers, oracle/oracle_ff.c "
                                         f"(same stage pipeline, scheduler, BFGS and term tables — like the GPU it evaluates ALL "
                                         f"N (N - 1) / 2 distance terms of a molecule and a dense inverse Hessian, it is a port of "
                                         f"this path, not RDKit's embedder; OpenMP over attempts / conformers on {threads} "
                                         f"threads), {c_embed + c_mmff:.1f} s"}
        if cpu_shared is not None:
            cpu_shared["cpu_baseline"] = out["cpu_baseline"]
    return out


def chembl_library():
    """The reference's benchmark molecules (benchmarks/data/chembl_10k.smi, kept as tests/golden/chembl_10k.smi) through the library's
    ingestion, explicit hydrogens from the valence model (benchmarks/bench_utils/molprep.py:21-55: AddHs), generic parameters
    (synthetic.graph_molecule): generated ONCE for both ChEMBL blocks.  Returns (library, seconds, atom counts of the whole file)."""
    from nvmolkit_amd import synthetic
    from nvmolkit_amd.fingerprints import SmilesSet

    path = ROOT / "tests" / "golden" / "chembl_10k.smi"
    t0 = time.perf_counter()
    s = SmilesSet.from_file(str(path))
    totals = np.array([int(s.graph(i)[0][:, 3].sum()) + int(s.n_atoms[i]) for i in range(len(s.status)) if s.status[i] == 0])
    library, _ = synthetic.smiles_file_library(path, max_atoms=None)
    return library, time.perf_counter() - t0, totals


def chembl_block(confs: int, mmff_iters: int, device, max_atoms: int | None = 128, cpu_seconds: float = 0.0, prepared=None,
                 cpu_shared: dict | None = None) -> dict:
    """BASELINE.json configs[2] on the reference's own molecules: the 10 000 SMILES of benchmarks/data/chembl_10k.smi through
    the library's ingestion (benchmarks/etkdg_bench.py:154-161 reads them with RDKit), ETKDG + MMFF94 on the REAL topologies with
    generic parameters.  ``max_atoms`` = 128: the drug-sized 89 % of the file; ``None``: EVERY molecule, to 1063 atoms — what the
    reference's benchmark feeds by default (benchmarks/etkdg_bench.py:193, --num_mols 0 = all) — whose peptides and macrocycles
    (10.5 % of the molecules, > 95 % of the work) are minimised by teams of workgroups (csrc/minimize_team.hip)."""
    library, t_library, totals = prepared if prepared is not None else chembl_library()
    if max_atoms is not None:
        library = [m for m in library if m["embed"]["n_atoms"] <= max_atoms]
    out = conformer_block(len(library), confs, mmff_iters, device, 1, 0, cpu_seconds, library, t_library,
                          data=f"topologies of tests/golden/chembl_10k.smi (the reference's benchmarks/data/chembl_10k.smi) with explicit "
                               f"hydrogens, {'every molecule' if max_atoms is None else f'at most {max_atoms} atoms'}; synthetic (generic) parameters: bounds from covalent radii, hybridisation "
                               f"and ring-size angles, cis / trans 1-4 windows, triangle smoothing; MMFF94-shaped terms with the same rest values",
                          resident_rerun=max_atoms is not None, cpu_sample_max_atoms=128,
                          pmc_file=None if max_atoms is not None else "pmc_hbm_traffic_conformers_chembl_whole_file.json",
                          cpu_shared=cpu_shared)
    out["atoms_histogram_of_the_whole_file"] = {"molecules": int(len(totals)), "mean": float(totals.mean()),
                                                "percentiles_1_10_50_90_99_max": [int(x) for x in np.percentile(totals, [1, 10, 50, 90, 99, 100])],
                                                "fraction_beyond_the_cut": float((totals > max_atoms).mean()) if max_atoms is not None else 0.0}
    sizes = np.array([m["embed"]["n_atoms"] for m in library])
    out["atoms_percentiles_of_the_run_5_25_50_75_95_max"] = [int(x) for x in np.percentile(sizes, [5, 25, 50, 75, 95, 100])]
    # the roofline of the <= 128-atom block divides the bytes the inverse-Hessian passes requested from HBM; the whole-file block has
    # a counter file of its own (tools/gpu_session.sh ... chembl_all_traffic), quoted while the kernel sources are those it was taken on
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-query", type=int, default=1_000_000, help="query fingerprints PER GPU")
    ap.add_argument("--n-ref", type=int, default=1_000_000, help="reference fingerprints (global)")
    ap.add_argument("--fp-bits", type=int, default=2048)
    ap.add_argument("--chunk-rows", type=int, default=16384,
                    help="query rows per launch (output block = rows x n_ref x 8 B = 131 GB at the default; with 128 or more tile "
                         "rows per launch an XCD owns 64 x 16 tiles of a supertile: 0.72 T pairs/s against 0.70 with 8192-row chunks)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (0 disables)")
    ap.add_argument("--path", choices=["mfma", "valu"], default="mfma",
                    help="mfma: FP4 matrix-core kernel on prepared sets (default); valu: v_bcnt popcount kernel")
    ap.add_argument("--butina-n", type=int, default=1_000_000,
                    help="also time fused Butina (cutoff 0.3) on this many rows (BASELINE.json configs[1]: 1M), reported "
                         "under 'secondary' (0 = skip)")
    ap.add_argument("--conformer-mols", type=int, default=10_000,
                    help="also time ETKDG (10 conformers) + MMFF94 optimise on this many synthetic drug-like molecules PER "
                         "GPU (BASELINE.json configs[2]: 10k; configs[3] when --gpus > 1), reported under 'secondary' (0 = skip)")
    ap.add_argument("--conformer-total", type=int, default=0,
                    help="STRONG scaling of the conformer block (BASELINE.json configs[3]: 100000): this many molecules for the whole "
                         "job, dealt over the ranks by cost (distributed.shard_molecules_by_cost); 0 = weak scaling, --conformer-mols per GPU")
    ap.add_argument("--chembl", type=int, default=1,
                    help="1: also run the conformer block on the ChEMBL topologies of tests/golden/chembl_10k.smi (single GPU), 0: skip")
    ap.add_argument("--chembl-all", type=int, default=1,
                    help="1: also run the conformer block on EVERY molecule of that file (to 1063 atoms; about two minutes), reported as "
                         "secondary.conformers_chembl_all")
    ap.add_argument("--conformer-confs", type=int, default=10)
    ap.add_argument("--cfg1", type=int, default=1, help="1: also run BASELINE configs[0] (10k SMILES -> Morgan -> 10k x 10k), 0: skip")
    ap.add_argument("--mmff-iters", type=int, default=200, help="MMFF maxIters (the reference benchmark's default)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    # NVMK_BENCH_SINGLE_RANK_COLLECTIVES=1 under torch.distributed.run with one rank walks the multi-rank code (RCCL init,
    # all-gather of the reference block, the max / sum all-reduces) on a one-GPU box; the numbers equal the plain run's
    distributed = world > 1 or os.environ.get("NVMK_BENCH_SINGLE_RANK_COLLECTIVES") == "1"
    library, t_library = None, 0.0
    if distributed and args.conformer_mols > 0:
        # the generator forks worker processes: under torchrun that happens before this process holds a HIP context and
        # RCCL's threads (the single-GPU run generates inside the conformer block, after the headline measurement)
        library, t_library = conformer_library(args.conformer_mols, world, rank, shared=world > 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)
        if world > 1 and args.conformer_mols > 0:
            library = share_library(library, rank)

    from nvmolkit_amd import _native

    lib = _native.lib()
    words = args.fp_bits // 32
    n_q, n_ref = args.n_query, args.n_ref
    chunk = min(args.chunk_rows, n_q)

    # inputs resident in HBM before the timed region
    queries = synth_fingerprints(n_q, words, device, SEED + 1 + rank)
    if distributed:
        from nvmolkit_amd.distributed import all_gather_rows, shard_bounds

        lo, hi = shard_bounds(n_ref, world, rank)
        ref_shard = synth_fingerprints(n_ref, words, device, SEED, row_range=(lo, hi)).contiguous()  # this rank's rows only
        ref_gathered = all_gather_rows(ref_shard, n_ref)
    else:
        ref_shard = None
        ref_gathered = synth_fingerprints(n_ref, words, device, SEED)
    out = torch.empty((chunk, n_ref), dtype=torch.float64, device=device)
    stream = torch.cuda.current_stream()
    sptr = int(stream.cuda_stream)
    n_launch = (n_q + chunk - 1) // chunk
    use_mfma = args.path == "mfma"
    ws_q = ws_r = None
    if use_mfma:
        assert chunk % 128 == 0 or chunk == n_q, "--chunk-rows must be a multiple of 128 on the mfma path"
        ws_q = torch.empty(lib.nvmk_fp4_workspace_bytes(n_q, args.fp_bits), dtype=torch.uint8, device=device)
        ws_r = torch.empty(lib.nvmk_fp4_workspace_bytes(n_ref, args.fp_bits), dtype=torch.uint8, device=device)
    else:
        _native.set_option("NVMK_SIM_PATH", "valu")

    def step() -> None:
        nonlocal ref_gathered
        if distributed:
            ref_gathered = all_gather_rows(ref_shard, n_ref)  # ONE RCCL all-gather over xGMI, 256 MB total at 1M x 2048 bit
        ref_ptr = ref_gathered.data_ptr()
        if use_mfma:
            # the bit -> FP4 expansion of both operands is part of the step (O(N + M), inside the timed region)
            _native.check(lib.nvmk_fp4_prepare(queries.data_ptr(), n_q, args.fp_bits, ws_q.data_ptr(), sptr))
            _native.check(lib.nvmk_fp4_prepare(ref_ptr, n_ref, args.fp_bits, ws_r.data_ptr(), sptr))
        for r0 in range(0, n_q, chunk):
            rows = min(chunk, n_q - r0)
            if use_mfma:
                rc = lib.nvmk_cross_similarity_prepared_f64(0, ws_q.data_ptr(), n_q, r0, rows, ws_r.data_ptr(), n_ref,
                                                            args.fp_bits, out.data_ptr(), n_ref, sptr)
            else:
                rc = lib.nvmk_cross_tanimoto_f64(queries.data_ptr() + r0 * words * 4, rows, ref_ptr, n_ref,
                                                 args.fp_bits, out.data_ptr(), n_ref, sptr)
            if rc != 0:
                _native.check(rc, "cross similarity launch")

    def barrier() -> None:
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)
    if distributed:
        t = torch.tensor([wall], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # spot check of the last block against the torch fp path is not needed: parity is the test-suite's job;
    # here only guard against a silently empty run.
    assert bool(torch.isfinite(out[:2, :1024]).all())

    pairs_per_step = float(n_q) * float(n_ref) * world
    value = pairs_per_step * args.steps / wall
    ms_per_step = wall * 1e3 / args.steps

    result = {
        "metric": "Tanimoto pairs/s (1M x 1M, 2048-bit)",
        "value": value,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "fp4(0/1) x fp4 -> f32 exact counts -> f64 ratio" if use_mfma else "u32 popcount -> f64",
        "data": "synthetic (seeded planted-cluster fingerprints, 2.3 % density, <=12 bit flips)",
        "config": {
            "workload": f"{n_q}x{n_ref} {args.fp_bits}-bit Tanimoto cross-similarity per GPU, dense float64 output "
                        f"in {n_launch} row-chunks of {chunk} rows (BASELINE.json configs[1]"
                        + ("; configs[4] sharding: queries sharded, reference all-gathered" if distributed else "") + ")",
            "n_query_per_gpu": n_q,
            "n_ref": n_ref,
            "fp_bits": args.fp_bits,
            "chunk_rows": chunk,
            "kernel_path": args.path,
            "parallelism": f"query-sharded x{world}" + (" + 1 RCCL all-gather/step" if distributed else ""),
        },
    }

    if rank == 0:
        # dominant kernel: cross_sim_tile_kernel.  Launches are back-to-back on one stream, so the mean
        # launch duration is the event-timed region / number of launches (rocprofv3 --kernel-trace --stats
        # under profiles/ reports the same average).
        launches = n_launch * args.steps
        avg_ms = gpu_ms / launches
        # algorithmic bytes of ONE STEP: every double written once, every reference row read once per launch, every query
        # row once — summed over the launches as they are (the last chunk is shorter), divided by the event time of the
        # same launches
        step_bytes = 0.0
        for r0 in range(0, n_q, chunk):
            rows = min(chunk, n_q - r0)
            step_bytes += 8.0 * rows * n_ref + (rows + n_ref) * (args.fp_bits / 8.0)
        algo_bytes = 8.0 * chunk * n_ref + (chunk + n_ref) * (args.fp_bits / 8.0)  # of a full-chunk launch (what the PMC file holds)
        achieved = step_bytes * args.steps / (gpu_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes committed under profiles/ (separate rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE runs of this same command; FETCH_SIZE doubled as the gfx950 note in MI355X_MICROARCH.md §HBM
        # prescribes).  Counters cannot be read inside this process, so the figure is reported only for the exact
        # workload it was collected on, else null.
        traffic, traffic_src = None, None
        for pmc in sorted((ROOT / "profiles").glob("r*/pmc_hbm_traffic_bench_launch.json"), reverse=True):
            c = json.loads(pmc.read_text())
            # quoted only for the exact workload AND kernel source it was collected on: a stale file is ignored
            if (use_mfma and chunk == c.get("chunk_rows", 8192) and n_ref == 1_000_000 and args.fp_bits == 2048
                    and c.get("kernel_source_sha256") == kernel_source_digest()):
                traffic = (2.0 * c["FETCH_SIZE"]["mean_KiB_per_full_launch"] + c["WRITE_SIZE"]["mean_KiB_per_full_launch"]) * 1024.0
                traffic_src = f"{pmc.relative_to(ROOT)}: (2 x FETCH_SIZE + WRITE_SIZE) KiB per full launch, kernel source hash matches"
                break
        result["roofline"] = {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "kernel": "nvmk::fp4::cross_sim_mfma_kernel<0>" if use_mfma else "nvmk::sim::cross_sim_tile_kernel<64,0>",
            "avg_launch_ms": avg_ms,
            "algorithmic_bytes_per_step": step_bytes,
            "algorithmic_bytes_per_full_launch": algo_bytes,
            "frac_of_measured_copy_bw_6300": achieved / 6300.0,
            "note": "achieved = algorithmic bytes of the whole step (all launches, the short last chunk included) x steps / "
                    "event-timed region; avg_launch_ms = that region / launches; on the mfma path the region also holds the "
                    "two O(N) fp4 expansions per step (<1 % of the time)",
        }
        if world == 1 and args.cpu_seconds > 0:
            ref_host = ref_gathered[: min(n_ref, 200_000)].cpu().numpy().view(np.uint32)
            result["cpu_baseline"] = cpu_baseline(ref_host, args.cpu_seconds)
        # The other half of BASELINE.json's metric line (Butina at threshold 0.7 = cutoff 0.3; mols/s of ETKDG + MMFF),
        # measured after and outside the timed region of the headline number, bounded to a few seconds.
        secondary = {}

        def guarded(name, fn, *fn_args):
            # a secondary block that fails is reported under its name; the headline line is still printed
            try:
                secondary[name] = fn(*fn_args)
            except Exception as exc:  # noqa: BLE001
                import gc
                import traceback

                gc.enable()  # (a block that failed inside its timed region left the collector off)
                traceback.print_exc()
                secondary[name] = {"error": f"{type(exc).__name__}: {exc}"}

        if world == 1 and args.butina_n > 0:
            guarded("fused_butina", butina_block, args.butina_n, words, device, args.cpu_seconds)
        if world == 1 and args.cfg1:
            out = None  # release the 65 GB output block first
            torch.cuda.empty_cache()
            guarded("cfg1_smiles_to_similarity", cfg1_block, device, args.cpu_seconds)
            guarded("morgan", morgan_block, device)
    else:
        secondary = {}
    if args.conformer_mols > 0:  # every rank takes part (configs[3]: molecules sharded, no collective on the data path)
        out = queries = ref_gathered = ref_shard = ws_q = ws_r = None  # noqa: F841  (release ~70 GB before the next block)
        torch.cuda.empty_cache()
        if not distributed:
            guarded("conformers", conformer_block, args.conformer_mols, args.conformer_confs, args.mmff_iters, device, world, rank,
                    args.cpu_seconds)
            prepared = None
            if args.chembl or args.chembl_all:
                try:
                    prepared = chembl_library()
                except Exception as exc:  # noqa: BLE001
                    secondary["conformers_chembl"] = {"error": f"{type(exc).__name__}: {exc}"}
            cpu_shared: dict = {}
            if args.chembl and prepared is not None:
                guarded("conformers_chembl", chembl_block, args.conformer_confs, args.mmff_iters, device, 128, args.cpu_seconds, prepared, cpu_shared)
            if args.chembl_all and prepared is not None:
                guarded("conformers_chembl_all", chembl_block, args.conformer_confs, args.mmff_iters, device, None, args.cpu_seconds, prepared, cpu_shared)
        else:  # ranks meet in collectives inside the block: an exception on one rank must end the job, not hang the others
            block = conformer_block(args.conformer_mols, args.conformer_confs, args.mmff_iters, device, world, rank, args.cpu_seconds,
                                    library, t_library, collectives=True, strong_total=args.conformer_total)
            if rank == 0:
                secondary["conformers"] = block
    if rank == 0:
        if secondary:
            result["secondary"] = secondary
        # the sources this line was measured on: tests/test_bench_contract.py refuses a committed line whose hashes differ from
        # those of the PMC files it quotes
        result["kernel_source_sha256"] = {"similarity": kernel_source_digest(), "conformers": conformer_source_digest(),
                                          "neighbour_count": kernel_source_digest(("similarity_mfma.hip", "count_panel.inc", "fp4.h", "tile_maps.h"))}
        # LAST key, compact: both halves of BASELINE.json's metric where a reader of the line's tail finds them
        def pick(block, *path):
            node = secondary.get(block, {})
            for key in path:
                node = node.get(key) if isinstance(node, dict) else None
            return round(node, 4) if isinstance(node, float) else node

        result["summary"] = {"tanimoto_pairs_per_s": round(value, 1), "tanimoto_frac": round(result["roofline"]["frac"], 4),
                             "conformers_mols_per_s": pick("conformers", "value"), "conformers_frac": pick("conformers", "roofline", "frac"),
                             "chembl128_mols_per_s": pick("conformers_chembl", "value"), "chembl128_frac": pick("conformers_chembl", "roofline", "frac"),
                             "chembl_all_mols_per_s": pick("conformers_chembl_all", "value"), "chembl_all_frac": pick("conformers_chembl_all", "roofline", "frac"),
                             "butina_s": pick("fused_butina", "seconds"), "butina_frac": pick("fused_butina", "roofline", "frac")}
        print(json.dumps(result))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
