// Workgroup -> tile maps of the matrix-core kernels (similarity_mfma.hip), as plain integer functions shared by the kernels,
// their launchers and a host-compiled test shim (tests/native/kernel_maps_host.hip, tests/test_tile_maps.py): every tile of a
// problem must be visited exactly once, whatever its shape.
#pragma once

#include <hip/hip_runtime.h>

namespace nvmk {
namespace maps {

constexpr unsigned kSuper = 64;  // supertile edge in tiles (both operand blocks of a supertile stay in L2 / Infinity Cache)

// Dense kernel.  blockIdx.y walks the supertiles (superM x 64 tiles, row-major over the problem); inside a supertile the
// map is XCD-aware: workgroup b runs on XCD b % 8 and every XCD has its own L2, so XCD x owns the (superM / 2) x 16-tile
// sub-block x (2 x 4 sub-blocks) and walks it with tile_n fastest — its L2 then holds 16 B tiles and the current A tile.
// superM is 128 when the launch has at least 128 tile rows (16 384-row chunks), else 64.
__host__ __device__ inline unsigned dense_super_m(const long long tilesM) { return tilesM >= 2 * kSuper ? 2 * kSuper : kSuper; }
__host__ __device__ inline void dense_tile(const unsigned bx, const unsigned by, const unsigned tilesN, const unsigned superM,
                                           unsigned& tile_m, unsigned& tile_n) {
  const unsigned superN = (tilesN + kSuper - 1) / kSuper;
  const unsigned sm     = by / superN;
  const unsigned sn     = by - sm * superN;
  const unsigned xcd    = bx & 7u;
  const unsigned local  = bx >> 3;  // 0 .. 8 superM - 1 inside the XCD's sub-block
  tile_m = sm * superM + (xcd >> 2) * (superM >> 1) + (local >> 4);
  tile_n = sn * kSuper + (xcd & 3u) * 16u + (local & 15u);
}

// Symmetric passes: only the supertiles on or above the diagonal are launched; sidx enumerates row sm = 0 .., columns
// sn = sm .. superN - 1.  Inverted with a double-precision square root and two correction loops (unsigned 64-bit products:
// past row 2 superN + 1 the product wraps and ends the search).  Returns false for the padding slots of the 2-D grid.
__host__ __device__ inline bool symmetric_supertile(const unsigned long long sidx, const unsigned superN, unsigned& sm, unsigned& sn) {
  const double S = static_cast<double>(superN);
  unsigned     r = static_cast<unsigned>((2.0 * S + 1.0 - sqrt((2.0 * S + 1.0) * (2.0 * S + 1.0) - 8.0 * static_cast<double>(sidx))) * 0.5);
  auto rowStart  = [&](const unsigned q) { return static_cast<unsigned long long>(q) * (2ull * superN - q + 1ull) / 2ull; };
  while (r > 0 && rowStart(r) > sidx) --r;
  while (rowStart(r + 1) <= sidx) ++r;
  sm = r;
  sn = r + static_cast<unsigned>(sidx - rowStart(r));
  return sm < superN;
}

// Count kernel: tile of workgroup bx inside supertile (sm, sn) of superH x superW tiles; a full 64 x 64 supertile is walked
// XCD-aware like the dense kernel's, a smaller one (skinny problems) row-major.
__host__ __device__ inline void count_tile(const unsigned bx, const unsigned sm, const unsigned sn, const unsigned superH,
                                           const unsigned superW, unsigned& tile_m, unsigned& tile_n) {
  if (superH == 64u && superW == 64u) {
    const unsigned xcd = bx & 7u, local = bx >> 3;
    tile_m = sm * 64u + (xcd >> 2) * 32u + (local >> 4);
    tile_n = sn * 64u + (xcd & 3u) * 16u + (local & 15u);
  } else {
    tile_m = sm * superH + bx / superW;
    tile_n = sn * superW + (bx - (bx / superW) * superW);
  }
}

// ---- row-panel count kernel (count_panel.inc) --------------------------------------------------------------------------------
// Panel of workgroup `block` (grid = 8 * slotsPerXcd; block & 7 = the XCD it runs on, block >> 3 = its slot there) in `round`:
// a round deals 8 groups of slotsPerXcd consecutive panels, one group per XCD, in boustrophedon order over the rounds (later
// panels have shorter sweeps: every XCD gets the same mix).  false: no panel for this workgroup in this round.
__host__ __device__ inline bool panel_of(const unsigned block, const unsigned round, const unsigned slotsPerXcd, const unsigned panelLo,
                                         const unsigned panelHi, unsigned& pnl) {
  const unsigned xcd = block & 7u, slot = block >> 3;
  const unsigned g   = (round & 1u) ? 7u - xcd : xcd;
  pnl                = panelLo + round * 8u * slotsPerXcd + g * slotsPerXcd + slot;
  return pnl < panelHi;
}
__host__ __device__ inline bool panel_round_exists(const unsigned round, const unsigned slotsPerXcd, const unsigned panelLo,
                                                   const unsigned panelHi) {
  return panelLo + round * 8u * slotsPerXcd < panelHi;
}
// Ring of chunk buffers: chunk q of a sweep (CH chunks per 64-column tile) lives in buffer q % ring; its DMA is issued `dist`
// steps before the step that reads it.  What a wave has issued after chunk (tile, ch)'s own data DMA when it waits for it: two DMA
// instructions for each of the dist - 1 chunks after it, plus the column-popcount DMA of every tile that starts among them.
__host__ __device__ constexpr int panel_pending_after(const int chunksPerTile, const int dist, const int ch) {
  int n = 0;
  for (int j = 1; j < dist; ++j) n += 2 + (((ch + j) % chunksPerTile) == 0 ? 1 : 0);
  return n;
}

}  // namespace maps
}  // namespace nvmk
