"""The workgroup -> tile maps of the matrix-core kernels and the inverse-Hessian layout of the BFGS kernels, tested on the REAL
code: tests/native/kernel_maps_host.hip compiles nvmolkit_amd/csrc/tile_maps.h (which cross_sim_mfma_kernel,
neighbor_count_mfma_kernel and their launchers include) and nvmolkit_amd/csrc/hess_pass.h for the host with hipcc — no GPU
needed.  Every tile of a problem must be visited exactly once, whatever its shape: XCD-aware walk inside a supertile, supertiles
of a rectangular problem, and the upper-triangular supertile enumeration of the symmetric passes (inverted with a
double-precision square root plus two correction loops)."""

import ctypes
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
SUPER = 64


@pytest.fixture(scope="module")
def maps(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("needs hipcc")
    out = tmp_path_factory.mktemp("maps") / "libkernel_maps.so"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
                    str(ROOT / "tests" / "native" / "kernel_maps_host.hip"), "-o", str(out)], check=True, capture_output=True)
    lib = ctypes.CDLL(str(out))
    u, up = ctypes.c_uint, ctypes.POINTER(ctypes.c_uint)
    lib.chk_dense_super_m.argtypes, lib.chk_dense_super_m.restype = [ctypes.c_longlong], u
    lib.chk_dense_tile.argtypes = [u, u, u, u, up, up]
    lib.chk_symmetric_supertile.argtypes, lib.chk_symmetric_supertile.restype = [ctypes.c_ulonglong, u, up, up], ctypes.c_int
    lib.chk_count_tile.argtypes = [u, u, u, u, u, up, up]
    lib.chk_hess_row_offset.argtypes, lib.chk_hess_row_offset.restype = [ctypes.c_int64], ctypes.c_int64
    lib.chk_hess_row_offset32.argtypes, lib.chk_hess_row_offset32.restype = [ctypes.c_int], ctypes.c_int
    lib.chk_lds_vector_doubles.argtypes, lib.chk_lds_vector_doubles.restype = [ctypes.c_int, ctypes.c_int64], ctypes.c_int64
    lib.chk_lds_hessian_doubles.argtypes, lib.chk_lds_hessian_doubles.restype = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64], ctypes.c_int64
    lib.chk_resident_rows.argtypes, lib.chk_resident_rows.restype = [ctypes.c_int, ctypes.c_int, ctypes.c_int64], ctypes.c_int
    lib.chk_tail_pad_doubles.restype = ctypes.c_int64
    lib.chk_panel_of.argtypes, lib.chk_panel_of.restype = [u, u, u, u, u, up], ctypes.c_int
    lib.chk_panel_round_exists.argtypes, lib.chk_panel_round_exists.restype = [u, u, u, u], ctypes.c_int
    lib.chk_panel_pending_after.argtypes, lib.chk_panel_pending_after.restype = [ctypes.c_int] * 3, ctypes.c_int
    return lib


def pair(fn, *args):
    a, b = ctypes.c_uint(0), ctypes.c_uint(0)
    r = fn(*args, ctypes.byref(a), ctypes.byref(b))
    return r, a.value, b.value


@pytest.mark.parametrize("tiles_m,tiles_n", [(1, 1), (5, 3), (64, 64), (65, 1), (127, 70), (128, 64), (129, 200), (300, 65)])
def test_dense_kernel_visits_every_tile_once(maps, tiles_m, tiles_n):
    """The launcher's grid (launch_dense: superM x 64 workgroups in x, one supertile per y) through the kernel's own map."""
    super_m = maps.chk_dense_super_m(tiles_m)
    assert super_m == (128 if tiles_m >= 128 else 64)
    supers = -(-tiles_m // super_m) * -(-tiles_n // SUPER)
    seen = np.zeros((tiles_m, tiles_n), dtype=np.int32)
    for by in range(supers):
        for bx in range(super_m * SUPER):
            _, tm, tn = pair(maps.chk_dense_tile, bx, by, tiles_n, super_m)
            if tm < tiles_m and tn < tiles_n:
                seen[tm, tn] += 1
    assert np.all(seen == 1)


def test_an_xcd_owns_one_sub_block_of_a_supertile(maps):
    """Workgroup b runs on XCD b % 8: its tiles form a (superM / 2) x 16 block, walked with tile_n fastest."""
    for super_m in (64, 128):
        for xcd in range(8):
            tiles = [pair(maps.chk_dense_tile, b, 0, 64, super_m)[1:] for b in range(xcd, super_m * SUPER, 8)]
            tm, tn = np.array(tiles).T
            assert tm.max() - tm.min() + 1 == super_m // 2 and tn.max() - tn.min() + 1 == 16
            assert len(set(tiles)) == len(tiles) == super_m * 8 and tiles[1] == (tiles[0][0], tiles[0][1] + 1)


@pytest.mark.parametrize("super_n", list(range(1, 40)) + [63, 64, 65, 123, 256, 511])
def test_symmetric_supertile_enumeration_is_the_upper_triangle(maps, super_n):
    want = [(a, b) for a in range(super_n) for b in range(a, super_n)]
    got = []
    for i in range(super_n * (super_n + 1) // 2):
        ok, sm, sn = pair(maps.chk_symmetric_supertile, i, super_n)
        assert ok == 1
        got.append((sm, sn))
    assert got == want
    # the padding slots of the 2-D supertile grid (grid y x z may overshoot) are refused
    for extra in range(super_n * (super_n + 1) // 2, super_n * (super_n + 1) // 2 + 3):
        assert pair(maps.chk_symmetric_supertile, extra, super_n)[0] == 0


@pytest.mark.parametrize("tiles", [1, 2, 17, 63, 64, 65, 130, 200])
def test_symmetric_count_pass_visits_the_upper_triangle_once(maps, tiles):
    """launch_counts' grid (superE x superW workgroups per supertile, upper-triangular supertiles) through the kernel's maps
    and its `tile_n < tile_m` exit."""
    super_w = min(tiles, SUPER)
    super_n = -(-tiles // super_w)
    seen = np.zeros((tiles, tiles), dtype=np.int32)
    for sidx in range(super_n * (super_n + 1) // 2):
        ok, sm, sn = pair(maps.chk_symmetric_supertile, sidx, super_n)
        assert ok
        for bx in range(SUPER * super_w):
            _, tm, tn = pair(maps.chk_count_tile, bx, sm, sn, SUPER, super_w)
            if tm < tiles and tn < tiles and tn >= tm:
                seen[tm, tn] += 1
    iu = np.triu(np.ones((tiles, tiles), dtype=bool))
    assert np.all(seen[iu] == 1) and np.all(seen[~iu] == 0)


# ---- packed inverse Hessian: row offsets, LDS layout, residency rule (hess_pass.h) ---------------------------------------

def test_packed_triangle_rows_are_contiguous_even_and_16_byte_aligned(maps):
    off = [maps.chk_hess_row_offset(r) for r in range(0, 2000)]
    assert off[0] == 0 and all(o % 2 == 0 for o in off)                     # every row starts on a 16-byte boundary
    for r in range(1, 1999):
        assert off[r + 1] - off[r] == r + (r & 1)                           # row r: r entries, padded to an even length
        assert maps.chk_hess_row_offset32(r) == off[r]
    assert maps.chk_hess_row_offset(46000) == 46000 * 46000 // 2 and maps.chk_hess_row_offset(46000) < 2**31   # the 32-bit bound of minimize.hip
    assert maps.chk_tail_pad_doubles() >= 256                               # unconditional 1 KB loads past the last rows


@pytest.mark.parametrize("threads", [64, 128, 256, 512])
def test_lds_layout_and_resident_rows(maps, threads):
    nw = threads // 64
    for n in (12, 60, 144, 176, 256, 655, 1320):
        vec = maps.chk_lds_vector_doubles(threads, n)
        assert vec == (11 + nw) * n + 8 * nw + 8                            # 10 vectors + (1 + NW) partial-sum slabs + reduction scratch
        for lds_bytes in (vec * 8, 19968, 40448, 80896, 163840):
            hld = maps.chk_lds_hessian_doubles(threads, lds_bytes // 8, n)
            assert hld == max(lds_bytes // 8 - vec, 0)
            rl = maps.chk_resident_rows(threads, n, hld)
            assert 0 <= rl <= n and maps.chk_hess_row_offset(rl) <= max(hld, 0) or rl == 0
            if rl < n:
                # one more row would not fit — up to the rounding the one-wave pass needs (whole groups of eight rows below 64,
                # even boundaries above: hess_packed / hess_range)
                slack = 8 if (threads == 64 and rl < 64) else (2 if threads == 64 else 1)
                assert maps.chk_hess_row_offset(rl + slack) > hld
            if threads == 64:
                assert rl == n or rl % (8 if rl < 64 else 2) == 0
    # the whole triangle resident when it fits
    assert maps.chk_resident_rows(threads, 40, maps.chk_hess_row_offset(40)) == 40


@pytest.mark.parametrize("slots,lo,hi", [(32, 0, 3907), (32, 0, 1), (32, 0, 256), (32, 0, 257), (38, 0, 1000), (32, 100, 612), (4, 3, 77), (1, 0, 9)])
def test_row_panel_hand_out_covers_every_panel_once(maps, slots, lo, hi):
    """count_panel.inc: the panels [lo, hi) over the rounds of a grid of 8 x slots workgroups — each exactly once, a group of
    `slots` consecutive panels on one XCD (block & 7), the XCD order reversed every other round, and with it every XCD's sum of
    sweep lengths (a panel's sweep is shorter the later the panel) within a round's worth of the mean."""
    seen, per_xcd, rounds = {}, [0] * 8, 0
    while maps.chk_panel_round_exists(rounds, slots, lo, hi):
        groups = {}
        for block in range(8 * slots):
            pnl = ctypes.c_uint(0)
            if maps.chk_panel_of(block, rounds, slots, lo, hi, ctypes.byref(pnl)):
                assert pnl.value not in seen
                seen[pnl.value] = (rounds, block)
                groups.setdefault(block & 7, []).append(pnl.value)
                per_xcd[block & 7] += hi - pnl.value          # the sweep's length, in panels
        for xcd, members in groups.items():
            assert sorted(members) == list(range(min(members), min(members) + len(members))), "a group is one run of consecutive panels"
        if rounds >= 1 and len(groups) == 8:
            first = [min(groups[x]) for x in range(8)]
            assert first == sorted(first, reverse=bool(rounds & 1))
        rounds += 1
    assert sorted(seen) == list(range(lo, hi)) and rounds == -(-(hi - lo) // (8 * slots))
    if rounds >= 4 and rounds % 2 == 0:
        assert max(per_xcd) - min(per_xcd) <= 8 * slots * slots      # the boustrophedon order evens the XCDs out


def test_row_panel_dma_accounting(maps):
    """The hand-written `s_waitcnt vmcnt(k)` of count_panel.inc: k = vector-memory operations issued after a chunk's own DMA.
    Counted here by replaying the issue order of a sweep (per chunk: the tile's popcount DMA if it is the tile's first chunk,
    then two data DMAs; `dist` chunks are issued before the first wait, then one per step after its wait)."""
    for chunks_per_tile in (2, 4, 8):
        for dist in (4, 10, 14):
            ops = []                                      # (chunk index, kind) in issue order
            def issue(q):
                if q % chunks_per_tile == 0:
                    ops.append((q, "popcounts"))
                ops.extend([(q, "data"), (q, "data")])
            for q in range(dist):
                issue(q)
            for q in range(40):
                last_own = max(i for i, (c, kind) in enumerate(ops) if c == q and kind == "data")
                assert maps.chk_panel_pending_after(chunks_per_tile, dist, q % chunks_per_tile) == len(ops) - 1 - last_own
                issue(q + dist)
