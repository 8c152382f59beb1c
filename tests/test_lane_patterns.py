"""Lane-level patterns of the kernels restated on 64-element arrays (no GPU):

* wave_sum4_transposed (nvmolkit_amd/csrc/hess_pass.h): four row sums reduced together with quad / row DPP steps and the two
  gfx950 permlane swaps — every lane must end with the TOTAL of row (lane & 3), and the summation tree must be the same for every
  row (the bitwise reproducibility of a minimisation rests on it);
* the half-wave variant of hess_packed (two rows per wave instruction, two row pairs per group);
* the XOR swizzle of the LDS operand chunks (similarity_mfma.hip Chunk<KCW>): the 16 lanes of a ds_read_b128 lane group land on
  16 different 16-byte slots of the 256-byte bank row, and the swizzle is an involution per row."""

import numpy as np
import pytest

LANES = np.arange(64)


def quad_perm(x, perm):  # DPP quad_perm: lane l reads lane (l & ~3) + perm[l & 3]
    return x[(LANES & ~3) + np.array(perm)[LANES & 3]]


def row_ror(x, k):  # DPP row_ror:k inside rows of 16 lanes: lane l reads lane (l - k) mod 16 of its row
    return x[(LANES & ~15) + ((LANES & 15) - k) % 16]


def swap_add(x, dist):  # v_permlane16_swap / v_permlane32_swap of a register with its own copy, then the add: x + x[lane ^ dist]
    return x + x[LANES ^ dist]


def wave_sum4_transposed(rs):
    odd, up = (LANES & 1) != 0, (LANES & 2) != 0
    ab = np.where(odd, rs[1], rs[0]) + quad_perm(np.where(odd, rs[0], rs[1]), [1, 0, 3, 2])
    cd = np.where(odd, rs[3], rs[2]) + quad_perm(np.where(odd, rs[2], rs[3]), [1, 0, 3, 2])
    x = np.where(up, cd, ab) + quad_perm(np.where(up, ab, cd), [2, 3, 0, 1])
    x = x + row_ror(x, 4)
    x = x + row_ror(x, 8)
    x = swap_add(x, 16)
    return swap_add(x, 32)


def test_four_row_reduction_gives_every_lane_its_rows_total():
    rng = np.random.default_rng(0)
    rs = rng.integers(-1000, 1000, size=(4, 64)).astype(np.float64)  # integers: exact sums, any order
    got = wave_sum4_transposed(rs)
    assert np.array_equal(got, rs.sum(axis=1)[LANES & 3])


def test_four_row_reduction_uses_one_summation_tree_for_every_row():
    """With floating-point inputs the result depends on the order of the additions: the same 64 values placed in row 0, 1, 2 or 3
    must give the same bits."""
    rng = np.random.default_rng(1)
    v = rng.normal(size=64) * 10.0 ** rng.integers(-8, 8, size=64)
    results = []
    for row in range(4):
        rs = rng.normal(size=(4, 64))
        rs[row] = v
        results.append(wave_sum4_transposed(rs)[row])
    assert len({float(r).hex() for r in results}) == 1


def test_half_wave_reduction_of_the_packed_rows():
    # hess_packed: lanes 0..31 hold row pair u of the lower row, lanes 32..63 of the upper one; two row pairs per group
    rng = np.random.default_rng(2)
    rs = rng.integers(-1000, 1000, size=(2, 64)).astype(np.float64)
    odd = (LANES & 1) != 0
    x = np.where(odd, rs[1], rs[0]) + quad_perm(np.where(odd, rs[0], rs[1]), [1, 0, 3, 2])
    x = x + quad_perm(x, [2, 3, 0, 1])
    x = x + row_ror(x, 4)
    x = x + row_ror(x, 8)
    x = swap_add(x, 16)
    for half in range(2):
        for u in range(2):
            want = rs[u, 32 * half:32 * half + 32].sum()
            writer = 32 * half + u  # (lane & 31) < 2 writes row r0 + 2 (lane & 1) + half
            assert x[writer] == want


@pytest.mark.parametrize("kcw", [4, 8, 16])
def test_lds_chunk_swizzle(kcw):
    def swz(row):
        return row & 15 if kcw == 16 else ((row >> 1) & 7 if kcw == 8 else (row >> 2) & 3)

    row_bytes = kcw * 16
    # a ds_read_b128 lane group = 16 consecutive rows reading the same logical slot
    for base in range(0, 128, 16):
        for slot in range(kcw):
            addr = np.array([(base + i) * row_bytes + ((slot ^ swz(base + i)) << 4) for i in range(16)])
            assert len(set((addr % 256) // 16)) == 16  # 16 different 16-byte slots of the 256-byte bank row
    for row in range(128):  # writer and reader apply the same term: an involution, and a permutation of the row's slots
        assert sorted(slot ^ swz(row) for slot in range(kcw)) == list(range(kcw))
        assert all((slot ^ swz(row)) ^ swz(row) == slot for slot in range(kcw))


def test_fp4_expansion_of_a_byte():
    """spread8 (similarity_mfma.hip prepare_kernel): bit k of a byte becomes 0x2 — FP4 e2m1 1.0 — in nibble k, 0x0 — 0.0 — otherwise,
    so that the FP4 dot product of two expanded rows is popcount(a & b)."""
    def spread8(x):
        x = (x | (x << 12)) & 0x000F000F
        x = (x | (x << 6)) & 0x03030303
        x = (x | (x << 3)) & 0x11111111
        return (x << 1) & 0xFFFFFFFF

    for byte in range(256):
        e = spread8(byte)
        assert [(e >> (4 * k)) & 0xF for k in range(8)] == [2 * ((byte >> k) & 1) for k in range(8)]
