"""Pins the CPU oracle (oracle/oracle_similarity.c) for the similarity / Butina path: hand-computed
vectors, the reference's numpy bit-unpack restatement, the reference's Butina property checkers and
its 10x10 known answer.  Runs without a GPU."""

import numpy as np
import pytest

import oracle
from tests import util


def test_handcomputed_vectors(golden_dir):
    g = np.load(golden_dir / "similarity_handcomputed.npz")
    words = g["words"]
    assert np.array_equal(oracle.cross_intersection(words), g["intersection"])
    assert np.array_equal(oracle.cross_similarity(words, metric=oracle.TANIMOTO), g["tanimoto"])  # bit-exact doubles
    np.testing.assert_allclose(oracle.cross_similarity(words, metric=oracle.COSINE), g["cosine"], rtol=0, atol=1e-15)
    # zero-union convention: 0 (reference src/similarity_kernels.cu:353-356: c / max(1, union))
    assert oracle.cross_similarity(words)[0, 0] == 0.0


@pytest.mark.parametrize("metric", [oracle.TANIMOTO, oracle.COSINE])
@pytest.mark.parametrize("key", ["random_128x64", "random_77x4", "clustered_300x32"])
def test_c_oracle_matches_numpy_restatement(golden_dir, key, metric):
    fps = np.load(golden_dir / "fingerprints_small.npz")[key]
    a, b = fps[: len(fps) // 2], fps[len(fps) // 3:]
    got = oracle.cross_similarity(a, b, metric=metric)
    ref = oracle.cross_similarity_numpy(a, b, metric=metric)
    if metric == oracle.TANIMOTO:
        assert np.array_equal(got, ref)  # integer counts + one IEEE division: identical
    else:
        np.testing.assert_allclose(got, ref, rtol=1e-15, atol=0)


def test_pack_unpack_roundtrip():
    rng = np.random.default_rng(1)
    bits = rng.random((10, 127)) < 0.5
    packed = oracle.pack_bits(bits)
    assert packed.shape == (10, 4)
    assert np.array_equal(oracle.unpack_bits(packed)[:, :127], bits)
    assert not oracle.unpack_bits(packed)[:, 127].any()


def test_threads_agree():
    fps = util.random_fingerprints(64, 64)
    assert np.array_equal(oracle.cross_similarity(fps, threads=1), oracle.cross_similarity(fps, threads=0))


def test_neighbor_counts_vs_numpy():
    x = util.clustered_fingerprints(150, 8, 6, max_flips=10, density=0.2)
    y = x[40:110]
    thr = np.float32(0.55)
    sim = oracle.cross_similarity_numpy(x, y).astype(np.float64)
    # float32 predicate: float(c)/float(u) >= thr
    inter = oracle.cross_intersection(x, y)
    pa = oracle.cross_intersection(x, x).diagonal()[:, None]
    pb = oracle.cross_intersection(y, y).diagonal()[None, :]
    u = pa + pb - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        pred = (inter.astype(np.float32) / u.astype(np.float32) >= thr) & (u > 0)
    want = pred.sum(axis=1).astype(np.int32)
    got = oracle.neighbor_counts(x, y, thr)
    assert np.array_equal(got, want)
    got2 = oracle.neighbor_counts(x, y, thr, sign=-1, counts=got.copy())
    assert not got2.any()
    assert sim.shape == (150, 70)


def test_butina_dense_known_answer(golden_dir):
    g = np.load(golden_dir / "butina_10x10.npz")
    labels, cent = oracle.butina_dense(g["dist"], float(g["cutoff"]))
    assert len(cent) == int(g["n_clusters"])
    assert sorted(np.flatnonzero(labels == 0)) == list(g["cluster0"]) and cent[0] == int(g["centroid0"])
    assert sorted(np.flatnonzero(labels == 1)) == list(g["cluster1"]) and cent[1] == int(g["centroid1"])
    for cid in range(2, 5):
        members = np.flatnonzero(labels == cid)
        assert len(members) == 1 and cent[cid] == members[0]


@pytest.mark.parametrize("n", [1, 10, 100, 400])
def test_butina_dense_properties(n):
    rng = np.random.default_rng(42)
    d = rng.random((n, n))
    d = np.abs(d - d.T)
    cutoff = 0.1
    labels, cent = oracle.butina_dense(d, cutoff)
    hit = d <= cutoff
    util.check_labels_valid(hit, labels)
    clusters = [tuple(np.flatnonzero(labels == c)) for c in range(len(cent))]
    util.check_greedy_butina(hit, clusters)
    for c, cl in zip(cent, clusters):
        assert c in cl


def test_butina_dense_edge_cases():
    n = 20
    labels, cent = oracle.butina_dense(np.zeros((n, n)), 0.5)  # one cluster
    assert (labels == 0).all() and len(cent) == 1
    d = np.ones((n, n)) - np.eye(n)
    labels, cent = oracle.butina_dense(d, 0.5)  # all singletons, ascending ids
    assert np.array_equal(labels, np.arange(n)) and np.array_equal(cent, np.arange(n))


@pytest.mark.parametrize("metric", [oracle.TANIMOTO, oracle.COSINE])
@pytest.mark.parametrize("key,cutoff", [("clustered_300x32", 0.4), ("clustered_500x64", 0.3), ("random_77x4", 0.5)])
def test_butina_fused_properties(golden_dir, key, cutoff, metric):
    x = np.load(golden_dir / "fingerprints_small.npz")[key]
    n = len(x)
    clusters, sizes, cent = oracle.butina_fused(x, cutoff, metric=metric)
    util.check_partition(clusters, n)
    assert sizes[0] == 0 and sizes[-1] == n and len(sizes) == len(clusters) + 1
    assert all(sizes[i + 1] - sizes[i] == len(clusters[i]) for i in range(len(clusters)))
    lens = [len(c) for c in clusters]
    assert all(lens[i] >= lens[i + 1] for i in range(len(lens) - 1))
    assert all(c[0] == z for c, z in zip(clusters, cent))
    # float32 neighbour predicate, as the oracle / reference kernels evaluate it
    inter = oracle.cross_intersection(x)
    pc = inter.diagonal()
    if metric == oracle.TANIMOTO:
        den = (pc[:, None] + pc[None, :] - inter).astype(np.float32)
    else:
        den = np.sqrt(pc[:, None].astype(np.float32) * pc[None, :].astype(np.float32))
    with np.errstate(divide="ignore", invalid="ignore"):
        hit = (inter.astype(np.float32) / den >= np.float32(1.0 - cutoff)) & (den > 0)
    util.check_greedy_butina(hit, clusters)


def test_butina_fused_edge_cases():
    one = util.random_fingerprints(1, 32)
    assert oracle.butina_fused(one, 0.5)[0] == [(0,)]
    same = np.repeat(util.random_fingerprints(1, 32), 50, axis=0)
    cl, sizes, cent = oracle.butina_fused(same, 0.5)
    assert len(cl) == 1 and set(cl[0]) == set(range(50)) and cl[0][0] == 49  # tie -> highest index
    rnd = util.random_fingerprints(50, 32, density=0.5)
    cl, _, _ = oracle.butina_fused(rnd, 0.001)
    assert len(cl) == 50 and all(len(c) == 1 for c in cl)
    zeros = np.zeros((5, 4), dtype=np.uint32)  # degree-0 rows become singletons
    cl, sizes, _ = oracle.butina_fused(zeros, 0.5)
    assert sorted(c[0] for c in cl) == list(range(5)) and sizes[-1] == 5


def test_division_shortcut_of_hip_epilogue_is_exact():
    """rcp_f32 seed (any of the three floats within 1 ulp) + one f64 Newton step + fma correction equals the IEEE
    quotient for every 0 <= c <= u <= 16384 (the device side is covered by
    tests/test_similarity_gpu.py::test_prefix_fingerprints_exhaust_all_ratios)."""
    assert oracle.check_newton_division(16384) == 0


def test_table_free_threshold_predicate_of_ring_kernel_is_exact():
    """c (1 + m) - s m >= 0 (> 0 for odd-significand thresholds), m = rounding boundary below thr, decides exactly like
    float(c) / float(s - c) >= thr for every 0 <= c < s <= 4096 — at round thresholds, at thresholds sitting exactly on
    a ratio and one ulp either side of it, at powers of two and at random floats in [2^-10, 1]."""
    rng = np.random.default_rng(7)
    thrs = [0.7, 0.3, 0.65, 0.5, 0.25, 1.0, 2.0 ** -10, 0.999, 1.0 / 3.0, 2.0 / 3.0]
    for c, u in [(1, 3), (2, 3), (7, 10), (13, 17), (100, 143), (511, 730), (1, 1024), (1023, 1024)]:
        t = np.float32(c) / np.float32(u)
        thrs += [float(t), float(np.nextafter(t, np.float32(0))), float(np.nextafter(t, np.float32(2)))]
    thrs += [float(np.float32(v)) for v in rng.uniform(2.0 ** -10, 1.0, size=12)]
    for thr in thrs:
        assert oracle.check_threshold_arith(thr, 4096 if thr in (0.7, 0.3) else 1500) == 0, thr


@pytest.mark.parametrize("words", [8, 16, 32, 48, 64, 128])
def test_vector_popcount_form_equals_the_scalar_loop(words):
    """bench.py's CPU baseline runs the port's AVX-512 VPOPCNTDQ form where the host has it (2 x 8 pairs at a time, fingerprints
    of whole 512-bit words): bit-identical to the scalar loop that pins the oracle — ragged sizes, empty rows."""
    lib = oracle.lib()
    a = util.random_fingerprints(77, words, 0.05, seed=5)
    b = util.random_fingerprints(203, words, 0.3, seed=6)
    a[0] = 0
    b[3] = 0
    try:
        lib.orc_set_scalar(1)
        want = oracle.cross_similarity(a, b, oracle.TANIMOTO)
        lib.orc_set_scalar(0)
        got = oracle.cross_similarity(a, b, oracle.TANIMOTO)
        one = oracle.cross_similarity(a, b, oracle.TANIMOTO, threads=1)
    finally:
        lib.orc_set_scalar(0)
    assert np.array_equal(got, want) and np.array_equal(one, want)
