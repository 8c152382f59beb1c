"""The workgroup -> tile maps of the matrix-core kernels (nvmolkit_amd/csrc/similarity_mfma.hip: cross_sim_mfma_kernel :196-203,
launch_dense :666-671; neighbor_count_mfma_kernel :397-426, launch_counts :686-699) restated in integer arithmetic: every tile of
a problem is visited exactly once, whatever its shape — XCD-aware walk inside a supertile, supertiles of a rectangular problem,
and the upper-triangular supertile enumeration of the symmetric passes (inverted with a double-precision square root plus two
correction loops).  No GPU."""

import math

import numpy as np
import pytest

SUPER = 64


def dense_tiles(tiles_m, tiles_n):
    super_m = 2 * SUPER if tiles_m >= 2 * SUPER else SUPER
    super_n = -(-tiles_n // SUPER)
    supers = -(-tiles_m // super_m) * super_n
    seen = np.zeros((tiles_m, tiles_n), dtype=np.int32)
    b = np.arange(super_m * SUPER)
    xcd, local = b & 7, b >> 3
    for y in range(supers):
        sm, sn = y // super_n, y % super_n
        tm = sm * super_m + (xcd >> 2) * (super_m >> 1) + (local >> 4)
        tn = sn * SUPER + (xcd & 3) * 16 + (local & 15)
        ok = (tm < tiles_m) & (tn < tiles_n)
        np.add.at(seen, (tm[ok], tn[ok]), 1)
    return seen


@pytest.mark.parametrize("tiles_m,tiles_n", [(1, 1), (5, 3), (64, 64), (65, 1), (127, 70), (128, 64), (129, 200), (300, 65), (7813, 5)])
def test_dense_kernel_visits_every_tile_once(tiles_m, tiles_n):
    assert np.all(dense_tiles(tiles_m, tiles_n) == 1)


def symmetric_supertile(sidx, super_n):
    """(sm, sn) of supertile index sidx in the row-major enumeration of the upper triangle (sm <= sn)."""
    s = float(super_n)
    disc = (2.0 * s + 1.0) * (2.0 * s + 1.0) - 8.0 * float(sidx)
    r = int((2.0 * s + 1.0 - math.sqrt(disc)) * 0.5) if disc >= 0.0 else 0  # (a NaN converts to 0 on the device)

    def row_start(q):  # unsigned 64-bit arithmetic, as on the device: past row 2 superN + 1 the product wraps and ends the search
        m = 1 << 64
        return (q * ((2 * super_n - q + 1) % m) % m) // 2

    while r > 0 and row_start(r) > sidx:
        r -= 1
    while row_start(r + 1) <= sidx:
        r += 1
    return r, r + (sidx - row_start(r))


@pytest.mark.parametrize("super_n", list(range(1, 40)) + [63, 64, 65, 123, 256, 511])
def test_symmetric_supertile_enumeration_is_the_upper_triangle(super_n):
    want = [(a, b) for a in range(super_n) for b in range(a, super_n)]
    got = [symmetric_supertile(i, super_n) for i in range(super_n * (super_n + 1) // 2)]
    assert got == want
    # the padding slots of the 2-D supertile grid (grid y x z may overshoot) land on a row >= superN and are dropped
    for extra in range(super_n * (super_n + 1) // 2, super_n * (super_n + 1) // 2 + 3):
        assert symmetric_supertile(extra, super_n)[0] >= super_n


def count_tiles(tiles, super_e=SUPER):
    """Tiles the symmetric count pass evaluates: those on or above the diagonal, once each."""
    super_w = min(tiles, super_e)
    super_n = -(-tiles // super_w)
    seen = np.zeros((tiles, tiles), dtype=np.int32)
    b = np.arange(super_e * super_w)
    for sidx in range(super_n * (super_n + 1) // 2):
        sm, sn = symmetric_supertile(sidx, super_n)
        if super_e == 64 and super_w == 64:
            xcd, local = b & 7, b >> 3
            tm = sm * 64 + (xcd >> 2) * 32 + (local >> 4)
            tn = sn * 64 + (xcd & 3) * 16 + (local & 15)
        else:
            tm = sm * super_e + b // super_w
            tn = sn * super_w + b % super_w
        ok = (tm < tiles) & (tn < tiles) & (tn >= tm)
        np.add.at(seen, (tm[ok], tn[ok]), 1)
    return seen


@pytest.mark.parametrize("tiles", [1, 2, 17, 63, 64, 65, 130, 200])
def test_symmetric_count_pass_visits_the_upper_triangle_once(tiles):
    seen = count_tiles(tiles)
    iu = np.triu(np.ones((tiles, tiles), dtype=bool))
    assert np.all(seen[iu] == 1) and np.all(seen[~iu] == 0)
