"""BASELINE.json configs[1] at its FULL size (1M x 1M, 2048 bit) through size-independent properties — the 8 TB
matrix has no oracle copy, so the dense kernel, the matrix-free neighbour-count kernel and fused Butina check each
other and sampled rows are compared with the CPU oracle."""

import numpy as np
import pytest
import torch

import oracle
from bench import SEED, synth_fingerprints
from nvmolkit_amd import _native
from nvmolkit_amd.clustering import fused_butina, update_neighbor_counts
from nvmolkit_amd.similarity import crossTanimotoSimilarity

pytestmark = pytest.mark.gpu

N = 1_000_000
THR = 0.7


@pytest.fixture(scope="module")
def data():
    return synth_fingerprints(N, 64, torch.device("cuda", 0), SEED)


def test_dense_rows_equal_neighbour_counts_at_full_size(data, native_lib, monkeypatch):
    """Every one of the 10^12 dense float64 entries is thresholded on the GPU and the per-row hit counts must equal what
    the matrix-free kernel (float32 predicate on exact integer counts, symmetric tile walk with column credits) reports;
    sampled rows must equal the CPU oracle bit for bit.  The planted-cluster data have popcounts < 130, so no ratio lies
    within 6e-5 of the threshold other than 7/10 itself, which both predicates accept."""
    _native.set_option("NVMK_SIM_PATH", "auto")
    x = data
    counts = torch.zeros(N, dtype=torch.int32, device="cuda")
    update_neighbor_counts(x, x, counts, THR)
    lib, sptr = native_lib, _native.stream_ptr(None)
    ws = torch.empty(lib.nvmk_fp4_workspace_bytes(N, 2048), dtype=torch.uint8, device="cuda")
    _native.check(lib.nvmk_fp4_prepare(x.data_ptr(), N, 2048, ws.data_ptr(), sptr))
    chunk = 8192
    out = torch.empty((chunk, N), dtype=torch.float64, device="cuda")
    thr64 = float(np.float32(THR))
    dense_counts = torch.empty(N, dtype=torch.int64, device="cuda")
    diag_ok = True
    sample_rows = {0, 4097, 123_456, 500_000, 999_999}
    for r0 in range(0, N, chunk):
        rows = min(chunk, N - r0)
        _native.check(lib.nvmk_cross_similarity_prepared_f64(0, ws.data_ptr(), N, r0, rows, ws.data_ptr(), N, 2048,
                                                            out.data_ptr(), N, sptr))
        block = out[:rows]
        dense_counts[r0:r0 + rows] = (block >= thr64).sum(dim=1)
        diag_ok = diag_ok and bool((block[torch.arange(rows), torch.arange(r0, r0 + rows)] == 1.0).all())
        for r in sample_rows:
            if r0 <= r < r0 + rows:
                want = oracle.cross_similarity(x[r:r + 1].cpu().numpy().view(np.uint32), x.cpu().numpy().view(np.uint32))
                assert np.array_equal(block[r - r0].cpu().numpy(), want[0]), r
    assert diag_ok                                                  # no empty fingerprints in this set
    assert torch.equal(dense_counts.to(torch.int32), counts)
    assert int(counts.min()) >= 1 and int(counts.max()) < 200       # planted clusters of ~50


def test_fused_butina_properties_at_full_size(data, monkeypatch):
    """Partition, greedy head, centroid-member similarity on sampled clusters, singleton tail — at 1M rows."""
    _native.set_option("NVMK_SIM_PATH", "auto")
    x = data
    cutoff = 1.0 - THR
    clusters, sizes, centroids = fused_butina(x, cutoff, return_centroids=True)
    assert sizes[0] == 0 and sizes[-1] == N and len(sizes) == len(clusters) + 1
    flat = np.fromiter((i for c in clusters for i in c), dtype=np.int64, count=N)
    assert np.array_equal(np.sort(flat), np.arange(N))              # a partition of the rows
    counts = torch.zeros(N, dtype=torch.int32, device="cuda")
    update_neighbor_counts(x, x, counts, THR)
    c = counts.cpu().numpy()
    # first cluster: the LAST row with the maximal degree and exactly its neighbours
    assert centroids[0] == int(np.nonzero(c == c.max())[0][-1]) and len(clusters[0]) == c.max()
    lens = np.array([len(k) for k in clusters])
    n_single = int((lens == 1).sum())
    head = lens[:len(lens) - n_single]
    assert (np.diff(head) <= 0).all()                               # greedy cluster sizes never grow
    rng = np.random.default_rng(5)
    thr32 = np.float32(THR)
    for k in rng.choice(len(head), size=40, replace=False):
        members = torch.tensor(clusters[k], device="cuda")
        assert clusters[k][0] == centroids[k]
        sim = crossTanimotoSimilarity(x[centroids[k]:centroids[k] + 1], x[members]).torch().cpu().numpy()[0]
        assert (sim.astype(np.float32) >= thr32).all()              # every member is a neighbour of its centroid
    tail = [k[0] for k in clusters[len(head):]]
    assert tail == sorted(tail)                                     # singleton tail ascending


def test_twenty_passes_of_the_row_panel_kernel_give_the_same_counts(data):
    """The row-panel count kernel synchronises its workgroups through global arrival counters (the group start barrier) and its
    waves through hand-placed waits: twenty all-pairs passes over the 1M rows must give the same neighbour counts every time, and
    those of the 128 x 128 tile kernel."""
    x = data
    with _native.options(NVMK_SIM_PATH="mfma", NVMK_COUNT_KERNEL="tile"):
        want = torch.zeros(N, dtype=torch.int32, device="cuda")
        update_neighbor_counts(x, x, want, THR)
    with _native.options(NVMK_SIM_PATH="mfma", NVMK_COUNT_KERNEL="panel"):
        for k in range(20):
            got = torch.zeros(N, dtype=torch.int32, device="cuda")
            update_neighbor_counts(x, x, got, THR)
            assert torch.equal(got, want), f"pass {k}: {int((got != want).sum())} rows differ"
