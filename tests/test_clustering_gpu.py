"""GPU parity: neighbour counting, fused Butina and dense Butina vs the CPU oracle and the
reference's property checkers (tests/test_butina.cpp:96-153, nvmolkit/tests/test_clustering.py)."""

import numpy as np
import pytest
import torch

import oracle
from nvmolkit_amd import _native
from nvmolkit_amd.clustering import butina, fused_butina, update_neighbor_counts
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["valu", "mfma", "mfma:dense", "mfma:serial", "mfma:table", "mfma:nosort", "mfma:panel", "mfma:panel-nosort", "auto"], autouse=True)
def sim_path(request):
    """Neighbour counting / fused Butina run on the v_bcnt kernels, on the FP4 matrix-core kernels (round loop on the
    sparse neighbour graph, as parallel sweeps over a bucket or — serial — one round at a time on a persistent workgroup;
    the dense round loop that streams the fingerprint matrix; the tile kernel
    with the threshold TABLE instead of the exact-arithmetic predicate it uses for thresholds in [2^-10, 1]; the all-pairs pass
    on the row-panel kernel, csrc/count_panel.inc, with and without the popcount-sorted copy) and on the
    library's automatic choice (the switches are set through nvmk_set_option: the library reads the environment once)."""
    path, _, variant = request.param.partition(":")
    with _native.options(NVMK_SIM_PATH=path, NVMK_BUTINA_ROUNDS=variant if variant in ("dense", "serial") else None,
                         NVMK_COUNT_THRESHOLD="table" if variant == "table" else None,
                         # fused Butina: all-pairs pass in input order (default: popcount-sorted copy with tile skipping)
                         NVMK_BUTINA_SORT="0" if variant in ("nosort", "panel-nosort") else None,
                         NVMK_COUNT_KERNEL="panel" if variant.startswith("panel") else None):
        yield request.param

METRICS = {"tanimoto": oracle.TANIMOTO, "cosine": oracle.COSINE}


def dev(words: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(words.view(np.int32)).cuda()


@pytest.mark.parametrize("metric", ["tanimoto", "cosine"])
@pytest.mark.parametrize("words", [4, 16, 32, 64, 128, 12, 3])
@pytest.mark.parametrize("thr", [0.0, 0.3, 0.55, 0.7, 1.0])
def test_neighbor_counts_match_oracle(metric, words, thr):
    x = util.clustered_fingerprints(391, words, 9, max_flips=10, density=0.15, seed=words)
    y = util.clustered_fingerprints(5000 if words == 64 else 263, words, 9, max_flips=10, density=0.15, seed=words)
    counts = torch.zeros(len(x), dtype=torch.int32, device="cuda")
    update_neighbor_counts(dev(x), dev(y), counts, thr, metric=metric)
    want = oracle.neighbor_counts(x, y, thr, metric=METRICS[metric])
    assert np.array_equal(counts.cpu().numpy(), want)
    update_neighbor_counts(dev(x), dev(y), counts, thr, subtract=True, metric=metric)
    assert not counts.any()


@pytest.mark.parametrize("metric", ["tanimoto", "cosine"])
@pytest.mark.parametrize("rows,words", [(391, 16), (1500, 64), (700, 32), (5000, 64), (263, 12)])
def test_neighbor_counts_of_a_set_against_itself(rows, words, metric):
    """x against x (the same buffer): the library evaluates the upper triangle only and credits both rows of a pair — the counts
    must be those of the full square, self pairs included (the first pass of the reference's fused Butina,
    nvmolkit/clustering.py:138-160, is this call)."""
    x = util.clustered_fingerprints(rows, words, 11, max_flips=10, density=0.15, seed=rows + words)
    dx = dev(x)
    counts = torch.zeros(rows, dtype=torch.int32, device="cuda")
    update_neighbor_counts(dx, dx, counts, 0.55, metric=metric)
    want = oracle.neighbor_counts(x, x, 0.55, metric=METRICS[metric])
    assert np.array_equal(counts.cpu().numpy(), want) and (want >= 1).all()
    update_neighbor_counts(dx, dx, counts, 0.55, subtract=True, metric=metric)
    assert not counts.any()


def test_neighbor_counts_threshold_boundaries():
    """Exact float32 boundary behaviour: pairs whose similarity equals the threshold are neighbours."""
    words = 4
    bits = np.zeros((4, 128), dtype=bool)
    bits[0, :10] = True           # |a| = 10
    bits[1, :7] = True            # c = 7, u = 10 -> 0.7
    bits[2, :3] = True            # 0.3
    bits[3, 5:15] = True          # c = 5, u = 15 -> 1/3
    x = util.pack_bits(bits)
    for thr in (np.float32(0.7), np.float32(7) / np.float32(10), np.nextafter(np.float32(0.7), np.float32(1)),
                np.float32(1) / np.float32(3), np.nextafter(np.float32(1) / np.float32(3), np.float32(0))):
        counts = torch.zeros(4, dtype=torch.int32, device="cuda")
        update_neighbor_counts(dev(x), dev(x), counts, float(thr))
        assert np.array_equal(counts.cpu().numpy(), oracle.neighbor_counts(x, x, thr)), thr


@pytest.mark.parametrize("metric", ["tanimoto", "cosine"])
@pytest.mark.parametrize("key,cutoff", [("clustered_300x32", 0.4), ("clustered_500x64", 0.3), ("clustered_500x64", 0.7),
                                        ("random_128x64", 0.5)])
def test_fused_butina_equals_oracle(golden_dir, key, cutoff, metric):
    x = np.load(golden_dir / "fingerprints_small.npz")[key]
    clusters, sizes, cent = fused_butina(dev(x), cutoff, return_centroids=True, metric=metric)
    w_clusters, w_sizes, w_cent = oracle.butina_fused(x, cutoff, metric=METRICS[metric])
    assert sizes == w_sizes and cent == w_cent and clusters == w_clusters
    util.check_partition(clusters, len(x))


@pytest.mark.parametrize("n,words,centres", [(2000, 64, 40), (3001, 32, 100), (4500, 64, 300)])
def test_fused_butina_larger_vs_oracle(n, words, centres):
    x = util.clustered_fingerprints(n, words, centres, seed=n)
    got = fused_butina(dev(x), 0.35, return_centroids=True)
    want = oracle.butina_fused(x, 0.35)
    assert got[1] == want[1] and got[2] == want[2] and got[0] == want[0]


def test_fused_butina_edge_cases():
    one = dev(util.random_fingerprints(1, 32))
    assert fused_butina(one, 0.5) == ([(0,)], [0, 1])
    same = dev(np.repeat(util.random_fingerprints(1, 32), 50, axis=0))
    cl, sizes = fused_butina(same, 0.5)
    assert len(cl) == 1 and set(cl[0]) == set(range(50)) and sizes == [0, 50]
    rnd = dev(util.random_fingerprints(50, 32, density=0.5))
    cl, _ = fused_butina(rnd, 0.001)
    assert len(cl) == 50 and all(len(c) == 1 for c in cl)
    zeros = torch.zeros((5, 4), dtype=torch.int32, device="cuda")
    cl, sizes = fused_butina(zeros, 0.5)
    assert sorted(c[0] for c in cl) == list(range(5)) and sizes[-1] == 5
    empty = torch.zeros((0, 4), dtype=torch.int32, device="cuda")
    assert fused_butina(empty, 0.5) == ([], [0])


@pytest.mark.parametrize("cutoff", [0.02, 0.1, 0.3])
def test_fused_butina_prefix_chain(cutoff):
    """Row i = the first i bits set (row 0 is empty): similarity min(i, j) / max(i, j) makes a banded neighbour graph
    full of equal degrees, rows that drop to degree 1 round after round, and one degree-0 row — the tie-break
    (last row wins), the singleton harvest and the end-game of the round loop all show up in the output."""
    n = 300
    idx = np.arange(n)
    words = 16
    full = idx[:, None] // 32 > np.arange(words)[None, :]
    part = idx[:, None] // 32 == np.arange(words)[None, :]
    rem = ((np.uint64(1) << (idx % 32).astype(np.uint64)) - np.uint64(1)).astype(np.uint32)
    x = np.where(full, np.uint32(0xFFFFFFFF), np.where(part, rem[:, None], np.uint32(0))).astype(np.uint32)
    got = fused_butina(dev(x), cutoff, return_centroids=True)
    want = oracle.butina_fused(x, cutoff)
    assert got[1] == want[1] and got[2] == want[2] and got[0] == want[0]


def test_fused_butina_dense_graph_falls_back():
    """More neighbour pairs than the edge buffer holds (128 per row): the sparse-graph loop hands over to the round
    loop that streams the fingerprint matrix, same result."""
    base = util.random_fingerprints(1, 64, density=0.3)
    x = np.repeat(base, 700, axis=0)
    x[np.arange(700), np.arange(700) % 64] ^= np.uint32(1) << (np.arange(700) % 32).astype(np.uint32)  # near-duplicates
    got = fused_butina(dev(x), 0.5, return_centroids=True)
    want = oracle.butina_fused(x, 0.5)
    assert got[1] == want[1] and got[2] == want[2] and got[0] == want[0]
    assert len(got[0]) == 1 and len(got[0][0]) == 700


def test_fused_butina_cluster_larger_than_the_staged_member_list(sim_path):
    """A 2600-member clique inside a sparse background: the graph (3.4 M pairs) still fits the edge buffer (128 per
    row), so the device-side round loop runs, and the first cluster overflows the 2048 members it hands from the
    extraction to the subtraction through LDS (the rest are re-read from the cluster list)."""
    if sim_path in ("valu", "mfma:dense"):
        pytest.skip("sparse round loop only")
    rng = np.random.default_rng(11)
    n, clique = 30000, 2600
    x = util.random_fingerprints(n, 16, density=0.25, seed=5)
    base = x[0].copy()
    x[:clique] = base
    flip = rng.integers(0, 16 * 32, size=clique)  # one bit flipped per clique member: all mutual neighbours at 0.5
    x[np.arange(clique), flip // 32] ^= (np.uint32(1) << (flip % 32).astype(np.uint32))
    perm = rng.permutation(n)
    x = np.ascontiguousarray(x[perm])
    got = fused_butina(dev(x), 0.5, return_centroids=True)
    want = oracle.butina_fused(x, 0.5)
    assert got[1] == want[1] and got[2] == want[2] and got[0] == want[0]
    assert len(got[0][0]) >= clique


def test_fused_butina_argument_validation():
    x = dev(util.random_fingerprints(10, 4))
    with pytest.raises(ValueError):
        fused_butina(x, 1.5)
    with pytest.raises(ValueError):
        fused_butina(x, 0.5, metric="dice")
    with pytest.raises(TypeError):
        fused_butina(x, 0.5, stream=3)
    with pytest.raises(ValueError):
        fused_butina(x.to(torch.int64), 0.5)
    with pytest.raises(ValueError):
        fused_butina(x.cpu(), 0.5)


@pytest.mark.parametrize("size,nl", [(s, n) for s in (1, 10, 100, 1000) for n in (8, 64, 128)])
def test_butina_dense_properties(size, nl):
    rng = np.random.default_rng(42)
    d = rng.random((size, size))
    d = np.abs(d - d.T)
    cutoff = 0.1
    res, cent = butina(torch.from_numpy(d).cuda(), cutoff, neighborlist_max_size=nl, return_centroids=True)
    labels = res.numpy()
    hit = d <= cutoff
    util.check_labels_valid(hit, labels)
    clusters = [tuple(np.flatnonzero(labels == c)) for c in range(labels.max() + 1)]
    util.check_greedy_butina(hit, clusters)
    w_labels, w_cent = oracle.butina_dense(d, cutoff)
    assert np.array_equal(labels, w_labels) and np.array_equal(cent.numpy(), w_cent)


def test_neighborlist_max_size_is_a_capacity_of_the_reference_kernels_not_part_of_the_result():
    """/root/reference/src/butina.cu:79-238: NeighborlistMaxSize is the capacity of the shared-memory neighbour lists the reference
    switches to once the largest remaining cluster fits them (kernels CountClusterSizeWithNeighborlist /
    attemptAssignClustersFromNeighborlist); before that point it runs the matrix kernels, and both produce the same greedy
    assignment.  Here the device rounds work on the whole graph at any cluster size, so the argument is validated and changes
    nothing: every allowed value gives the oracle's clusters on a set whose clusters are both larger and smaller than every
    capacity (planted clusters of 3 ... 300 members)."""
    rng = np.random.default_rng(7)
    sizes = [300, 150, 130, 70, 40, 20, 9, 3] + [1] * 30
    centre = np.repeat(np.arange(len(sizes)), sizes)
    n = len(centre)
    d = np.where(centre[:, None] == centre[None, :], rng.uniform(0.0, 0.09, (n, n)), rng.uniform(0.3, 1.0, (n, n)))
    d = np.minimum(d, d.T)
    np.fill_diagonal(d, 0.0)
    want_labels, want_cent = oracle.butina_dense(d, 0.1)
    assert np.bincount(want_labels).max() == 300
    x = torch.from_numpy(d).cuda()
    for nl in (8, 16, 24, 32, 64, 128):
        res, cent = butina(x, 0.1, neighborlist_max_size=nl, return_centroids=True)
        assert np.array_equal(res.numpy(), want_labels) and np.array_equal(cent.numpy(), want_cent), nl


def test_butina_dense_known_answer(golden_dir):
    g = np.load(golden_dir / "butina_10x10.npz")
    res, cent = butina(torch.from_numpy(g["dist"]).cuda(), float(g["cutoff"]), return_centroids=True)
    labels, cent = res.numpy(), cent.numpy()
    assert len(cent) == 5
    assert sorted(np.flatnonzero(labels == 0)) == [0, 1, 2, 3] and cent[0] == 0
    assert sorted(np.flatnonzero(labels == 1)) == [4, 5, 6] and cent[1] == 4
    for cid in range(2, 5):
        m = np.flatnonzero(labels == cid)
        assert len(m) == 1 and cent[cid] == m[0]


def test_butina_dense_edges_and_validation():
    n = 50
    assert (butina(torch.zeros((n, n), dtype=torch.float64, device="cuda"), 0.5).numpy() == 0).all()
    d = torch.ones((n, n), dtype=torch.float64, device="cuda") - torch.eye(n, dtype=torch.float64, device="cuda")
    assert np.array_equal(butina(d, 0.5).numpy(), np.arange(n))
    with pytest.raises(ValueError):
        butina(d, 0.5, neighborlist_max_size=7)
    with pytest.raises(TypeError):
        butina(d, 0.5, stream=1)
    with pytest.raises(ValueError):
        butina(d[:, :10], 0.5)


def test_fused_matches_dense_on_same_data():
    """Consistency across the two Butina paths: same adjacency -> same greedy clusters."""
    x = util.clustered_fingerprints(600, 64, 15, seed=5)
    cutoff = 0.4
    clusters, _ = fused_butina(dev(x), cutoff)
    inter = oracle.cross_intersection(x)
    pc = inter.diagonal()
    den = (pc[:, None] + pc[None, :] - inter).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        hit = (inter.astype(np.float32) / den >= np.float32(1.0 - cutoff)) & (den > 0)
    util.check_greedy_butina(hit, clusters)


@pytest.mark.parametrize("n_shards", [1, 2, 5])
@pytest.mark.parametrize("n", [700, 4000])
def test_sharded_pairs_assemble_to_the_fused_result(n, n_shards):
    """nvmk_butina_pairs per shard (bands of tile rows of the symmetric pass) -> summed degrees + concatenated pairs ->
    nvmk_butina_from_pairs == nvmk_butina_fused == the oracle (the single-GPU half of SURVEY.md 8(e) row 3; the exchange
    itself is tests/test_distributed_cpu.py)."""
    from nvmolkit_amd.distributed import butina_from_pairs_gpu, butina_pairs_gpu

    x = util.clustered_fingerprints(n, 64, max(4, n // 60), seed=n)
    d = dev(x)
    parts = [butina_pairs_gpu(d, 0.35, s, n_shards) for s in range(n_shards)]
    counts = sum(p[0] for p in parts)
    pairs = torch.cat([p[1] for p in parts])
    assert np.array_equal(counts.cpu().numpy(), oracle.neighbor_counts(x, x, np.float32(0.65)))
    pp = pairs.cpu().numpy()
    lo_, hi_ = pp.min(1).astype(np.int64), pp.max(1).astype(np.int64)
    assert np.all(lo_ != hi_) and len(np.unique(lo_ * n + hi_)) == len(pp)          # every unordered pair once, no self pairs
    got = butina_from_pairs_gpu(n, counts, pairs)
    assert got == oracle.butina_fused(x, 0.35)
    assert got == fused_butina(d, 0.35, return_centroids=True)


def test_from_pairs_refuses_a_graph_it_cannot_trust_and_pairs_retry_with_the_reported_capacity():
    """nvmk_butina_from_pairs checks the caller's pair list (indices inside [0, N), degrees = 1 + pairs of the row)
    before the CSR fill trusts it; butina_pairs_gpu asks again with the reported number of pairs when its buffer was too small."""
    from nvmolkit_amd.distributed import butina_from_pairs_gpu, butina_pairs_gpu

    n = 900
    x = util.clustered_fingerprints(n, 64, 6, seed=3)
    d = dev(x)
    counts, pairs = butina_pairs_gpu(d, 0.35, 0, 1)
    small_counts, small_pairs = butina_pairs_gpu(d, 0.35, 0, 1, capacity=16)   # far too small: one retry
    assert torch.equal(small_counts, counts) and small_pairs.shape == pairs.shape
    want = butina_from_pairs_gpu(n, counts, pairs)
    assert want == butina_from_pairs_gpu(n, small_counts, small_pairs)
    bad = pairs.clone()
    bad[0, 1] = n + 5
    with pytest.raises(ValueError, match="outside"):
        butina_from_pairs_gpu(n, counts, bad)
    with pytest.raises(ValueError, match="degree"):
        butina_from_pairs_gpu(n, counts + 1, pairs)
    with pytest.raises(ValueError, match="degree"):
        butina_from_pairs_gpu(n, counts, pairs[:-3])
