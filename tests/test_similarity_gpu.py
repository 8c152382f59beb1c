"""GPU parity: dense cross-similarity through the C ABI / Python API vs the CPU oracle.
Modelled on the reference's tests/test_similarity.cpp:190-250,383-445 and
nvmolkit/tests/test_similarity.py (RDKit replaced by the pinned oracle)."""

import numpy as np
import pytest
import torch

import oracle
from nvmolkit_amd import _native
from nvmolkit_amd.similarity import (crossCosineSimilarity, crossCosineSimilarityMemoryConstrained,
                                     crossTanimotoSimilarity, crossTanimotoSimilarityMemoryConstrained)
from nvmolkit_amd.types import AsyncGpuResult
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["valu", "mfma", "auto"], autouse=True)
def sim_path(request):
    """Every parity case runs on the VALU popcount kernel, on the FP4 matrix-core kernel and on the library's automatic
    choice (NVMK_SIM_PATH, set through nvmk_set_option: the library reads the environment once per process)."""
    with _native.options(NVMK_SIM_PATH=request.param):
        yield request.param

FUNCS = {"tanimoto": (crossTanimotoSimilarity, oracle.TANIMOTO), "cosine": (crossCosineSimilarity, oracle.COSINE)}


def dev(words: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(words.view(np.int32)).cuda()


def assert_matches(got: torch.Tensor, want: np.ndarray, metric: str):
    g = got.cpu().numpy()
    assert g.dtype == np.float64 and g.shape == want.shape
    if metric == "tanimoto":
        # integer counts + one correctly rounded double division on both sides: bit-exact
        assert np.array_equal(g, want), f"max |diff| = {np.abs(g - want).max()}"
    else:
        # double sqrt + division; both IEEE-correct on CPU and gfx950 -> tolerate 1 ulp only
        np.testing.assert_allclose(g, want, rtol=2.3e-16, atol=0)


def test_handcomputed(golden_dir):
    g = np.load(golden_dir / "similarity_handcomputed.npz")
    x = dev(g["words"])
    assert np.array_equal(crossTanimotoSimilarity(x).numpy(), g["tanimoto"])
    np.testing.assert_allclose(crossCosineSimilarity(x).numpy(), g["cosine"], rtol=0, atol=1e-15)


@pytest.mark.parametrize("metric", ["tanimoto", "cosine"])
@pytest.mark.parametrize("words", [4, 8, 16, 32, 64, 128, 96, 3, 1, 5])
@pytest.mark.parametrize("nm", [(1, 1), (1, 20), (20, 1), (29, 29), (100, 10), (130, 257), (257, 129)])
def test_cross_similarity_shapes(metric, words, nm):
    n, m = nm
    fn, mid = FUNCS[metric]
    a = util.random_fingerprints(n, words, density=0.2, seed=n * 1000 + words)
    b = util.random_fingerprints(m, words, density=0.1, seed=m * 1000 + words + 7)
    got = fn(dev(a), dev(b)).torch()
    torch.cuda.synchronize()
    assert_matches(got, oracle.cross_similarity(a, b, metric=mid), metric)


@pytest.mark.parametrize("metric", ["tanimoto", "cosine"])
def test_self_similarity_and_async_result(metric):
    fn, mid = FUNCS[metric]
    a = util.clustered_fingerprints(700, 64, 20)
    x = dev(a)
    want = oracle.cross_similarity(a, metric=mid)
    got = fn(x)
    assert isinstance(got, AsyncGpuResult) and got.torch().device.type == "cuda"
    assert_matches(got.torch(), want, metric)
    # AsyncGpuResult accepted as input, like the reference (nvmolkit/tests/test_similarity.py:107-109)
    assert_matches(fn(AsyncGpuResult(x)).torch(), want, metric)
    diag = got.torch().diagonal().cpu().numpy()
    assert np.all(diag == 1.0)


def test_zero_fingerprints_and_dense_rows():
    a = np.zeros((5, 64), dtype=np.uint32)
    a[1] = 0xFFFFFFFF
    a[2, 0] = 1
    got_t = crossTanimotoSimilarity(dev(a)).numpy()
    got_c = crossCosineSimilarity(dev(a)).numpy()
    assert got_t[0, 0] == 0.0 and got_c[0, 0] == 0.0 and got_t[0, 1] == 0.0
    assert got_t[1, 1] == 1.0 and got_t[1, 2] == 1.0 / 2048.0
    assert np.array_equal(got_t, oracle.cross_similarity(a))


def test_odd_leading_dimension_and_unaligned_views():
    a = util.random_fingerprints(131, 64, seed=3)
    b = util.random_fingerprints(127, 64, seed=4)  # odd M -> scalar store path
    assert_matches(crossTanimotoSimilarity(dev(a), dev(b)).torch(), oracle.cross_similarity(a, b), "tanimoto")
    # non-contiguous input is made contiguous by the Python layer
    big = dev(util.random_fingerprints(64, 128, seed=5))
    view = big[:, :64]
    want = oracle.cross_similarity(view.cpu().numpy().view(np.uint32).copy())
    assert_matches(crossTanimotoSimilarity(view).torch(), want, "tanimoto")


@pytest.mark.parametrize("fn", [crossTanimotoSimilarity, crossCosineSimilarity])
def test_fp_size_mismatch_raises(fn):
    a = dev(util.random_fingerprints(10, 4))
    b = dev(util.random_fingerprints(10, 8))
    with pytest.raises(ValueError):
        fn(a, b)


@pytest.mark.parametrize("fn", [crossTanimotoSimilarity, crossCosineSimilarity])
def test_bad_stream_type_raises(fn):
    a = dev(util.random_fingerprints(4, 4))
    with pytest.raises(TypeError):
        fn(a, stream=7)


def test_explicit_stream():
    a = util.random_fingerprints(300, 64, seed=11)
    s = torch.cuda.Stream()
    x = dev(a)
    torch.cuda.synchronize()
    out = crossTanimotoSimilarity(x, stream=s)
    s.synchronize()
    assert np.array_equal(out.torch().cpu().numpy(), oracle.cross_similarity(a))


@pytest.mark.parametrize("metric", ["tanimoto", "cosine"])
def test_memory_constrained_paths(metric):
    mid = FUNCS[metric][1]
    fn = crossTanimotoSimilarityMemoryConstrained if metric == "tanimoto" else crossCosineSimilarityMemoryConstrained
    a = util.random_fingerprints(300, 64, seed=21)
    b = util.random_fingerprints(170, 64, seed=22)
    want = oracle.cross_similarity(a, b, metric=mid)
    one_shot = fn(dev(a), dev(b))
    assert isinstance(one_shot, np.ndarray)
    # forced segmentation (reference: tests/test_similarity.cpp "forced-segmentation" cases):
    # allow 2 buffers x 64 rows -> several chunks on two streams
    chunked = fn(dev(a), dev(b), max_device_memory_bytes=int(2 * 64 * 170 * 8 / 0.9) + 64)
    for got in (one_shot, chunked):
        if metric == "tanimoto":
            assert np.array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, rtol=2.3e-16, atol=0)
    with pytest.raises(RuntimeError):
        fn(dev(a), dev(b), max_device_memory_bytes=1024)  # 32 rows do not fit


@pytest.mark.parametrize("metric", ["tanimoto", "cosine"])
def test_memory_constrained_chunks_on_operands_expanded_once(metric):
    """Several chunks large enough for the matrix-core path (reference: src/similarity.cpp:153-232, the chunk loop): both operands are
    expanded to FP4 once and every chunk is a launch on the prepared sets at a row offset that is a multiple of 128 — 1000 x 20 000
    with room for two buffers of 300-odd rows: chunks of 256 rows, the last one short.  Bit for bit the oracle's matrix."""
    mid = FUNCS[metric][1]
    fn = crossTanimotoSimilarityMemoryConstrained if metric == "tanimoto" else crossCosineSimilarityMemoryConstrained
    a = util.random_fingerprints(1000, 8, seed=31)
    b = util.random_fingerprints(20000, 8, seed=32)
    want = oracle.cross_similarity(a, b, metric=mid)
    got = fn(dev(a), dev(b), max_device_memory_bytes=int(2 * 300 * 20000 * 8 / 0.9))
    if metric == "tanimoto":
        assert np.array_equal(got, want)
    else:
        np.testing.assert_allclose(got, want, rtol=2.3e-16, atol=0)


@pytest.mark.parametrize("words", [32, 64, 128])
def test_prefix_fingerprints_exhaust_all_ratios(words):
    """Row i = the first i bits set: the (F+1) x (F+1) matrix contains EVERY ratio c / u with 0 <= c <= u <= F
    (c = min(i, j), u = max(i, j)), so the matrix-core kernel's division shortcut (Newton step from
    v_rcp_f32) is checked exhaustively against the IEEE quotient."""
    bits = words * 32
    n = bits + 1
    idx = np.arange(n)
    full = idx[:, None] // 32 > np.arange(words)[None, :]
    part = idx[:, None] // 32 == np.arange(words)[None, :]
    rem = (np.uint64(1) << (idx % 32).astype(np.uint64)) - np.uint64(1)
    fp = np.where(full, np.uint32(0xFFFFFFFF), np.where(part, rem[:, None].astype(np.uint32), np.uint32(0))).astype(np.uint32)
    assert np.array_equal(np.unpackbits(fp.view(np.uint8), axis=1).sum(axis=1), idx)
    got = crossTanimotoSimilarity(dev(fp)).torch().cpu().numpy()
    lo = np.minimum(idx[:, None], idx[None, :]).astype(np.float64)
    hi = np.maximum(idx[:, None], idx[None, :]).astype(np.float64)
    want = np.divide(lo, hi, out=np.zeros_like(lo), where=hi > 0)
    assert np.array_equal(got, want)


def test_full_size_properties_2048bit():
    """BASELINE-size width at a GPU-sized N: size-independent properties instead of an oracle matrix."""
    n = 8192
    a = util.clustered_fingerprints(n, 64, 200, shuffle=True, seed=99)
    x = dev(a)
    t = crossTanimotoSimilarity(x).torch()
    assert torch.equal(t, t.T)                       # symmetry, bit-exact
    assert bool((t.diagonal() == 1.0).all())         # non-empty rows
    assert bool(((t >= 0) & (t <= 1)).all())
    # checksum of checksums against the oracle on a row sample
    rows = np.arange(0, n, 97)
    want = oracle.cross_similarity(a[rows], a)
    assert np.array_equal(t[torch.from_numpy(rows).cuda()].cpu().numpy(), want)


def test_prepared_set_api_row_chunks(native_lib):
    """nvmk_fp4_prepare + nvmk_cross_similarity_prepared_f64 on row chunks == one-shot oracle matrix."""
    from nvmolkit_amd import _native

    a = util.clustered_fingerprints(1000, 64, 30, seed=31)
    b = util.clustered_fingerprints(777, 64, 30, seed=32)
    xa, xb = dev(a), dev(b)
    wa = torch.empty(native_lib.nvmk_fp4_workspace_bytes(len(a), 2048), dtype=torch.uint8, device="cuda")
    wb = torch.empty(native_lib.nvmk_fp4_workspace_bytes(len(b), 2048), dtype=torch.uint8, device="cuda")
    sptr = _native.stream_ptr(None)
    _native.check(native_lib.nvmk_fp4_prepare(xa.data_ptr(), len(a), 2048, wa.data_ptr(), sptr))
    _native.check(native_lib.nvmk_fp4_prepare(xb.data_ptr(), len(b), 2048, wb.data_ptr(), sptr))
    want = oracle.cross_similarity(a, b)
    for chunk in (128, 384):
        out = torch.full((len(a), len(b)), -1.0, dtype=torch.float64, device="cuda")
        for r0 in range(0, len(a), chunk):
            rows = min(chunk, len(a) - r0)
            _native.check(native_lib.nvmk_cross_similarity_prepared_f64(0, wa.data_ptr(), len(a), r0, rows, wb.data_ptr(),
                                                                        len(b), 2048, out[r0].data_ptr(), len(b), sptr))
        assert np.array_equal(out.cpu().numpy(), want)
    with pytest.raises(ValueError):
        _native.check(native_lib.nvmk_cross_similarity_prepared_f64(0, wa.data_ptr(), len(a), 5, 10, wb.data_ptr(), len(b),
                                                                    2048, out.data_ptr(), len(b), sptr))
