// FP4 ("prepared") fingerprint sets for the matrix-core similarity kernels.
//
// popcount(a & b) over K bits is the dot product of two 0/1 vectors.  The reference runs it on
// NVIDIA's 1-bit tensor-core MMA (mma.sync.m16n8k256.b1.and.popc, src/utils/macros_ptx.cuh:137-211);
// CDNA4 has no 1-bit MFMA, but its densest format, FP4 e2m1, represents 0.0 (0x0) and 1.0 (0x2)
// exactly, products are 0/1 and the f32 accumulator is exact for counts < 2^24.  So each fingerprint
// bit is expanded ONCE to a 4-bit nibble (O(N) work, 4x the packed bytes) and the O(N*M*K) work runs
// on v_mfma_scale_f32_32x32x64_f8f6f4 at ~8.2 PFLOP/s measured (tools/probe_mfma_fp4.hip: bit-exact,
// 2.0 T pairs/s at 2048 bits vs 0.32 T pairs/s for the v_bcnt VALU ceiling, profiles/r01_valu_dense).
//
// Workspace layout of a prepared set of n fingerprints of W words (fp_bits = 32 W):
//   [ int32 popc[nPad] ][ pad to 256 B ][ uint4 rows[nPad][Wp] ]
//   nPad = round_up(n, 768) (zero rows), Wp = round_up(W, 16) (zero words): word w of a row becomes the
//   16 bytes rows[row][w] = 32 nibbles, nibble k = 0x2 if bit k of the word is set.
#pragma once

#include "common.h"

namespace nvmk {
namespace fp4 {

constexpr int ROW_PAD   = 128;  // rows per workgroup tile of the 128 x 128 kernels; alignment of row chunks
constexpr int ROW_ALLOC = 768;  // prepared sets are zero-padded to a multiple of this (a multiple of the 128-row tiles; 768 also
                                // covers the 192- / 256-row tiles of the kernels under tools/experiments)
constexpr int WORD_PAD = 16;   // words per LDS K-chunk

struct Layout {
  int64_t n;       // valid rows
  int64_t nPad;    // rows allocated (multiple of ROW_ALLOC)
  int     W;       // packed words per fingerprint
  int     Wp;      // expanded words per row (multiple of WORD_PAD)
  size_t  rowsOffset;
  size_t  bytes;
};

inline Layout layout(int64_t n, int fpBits) {
  Layout L;
  L.n                   = n;
  L.nPad                = (n + ROW_ALLOC - 1) / ROW_ALLOC * ROW_ALLOC;
  L.W                   = fpBits / 32;
  L.Wp                  = (L.W + WORD_PAD - 1) / WORD_PAD * WORD_PAD;
  const size_t popBytes = static_cast<size_t>(L.nPad) * sizeof(int32_t);
  L.rowsOffset          = (popBytes + 255) / 256 * 256;
  L.bytes               = L.rowsOffset + static_cast<size_t>(L.nPad) * static_cast<size_t>(L.Wp) * 16;
  return L;
}

struct Prepared {
  const int32_t* popc;
  const uint4*   rows;
  Layout         L;
};

inline Prepared view(const void* ws, int64_t n, int fpBits) {
  Prepared P;
  P.L    = layout(n, fpBits);
  P.popc = static_cast<const int32_t*>(ws);
  P.rows = reinterpret_cast<const uint4*>(static_cast<const char*>(ws) + P.L.rowsOffset);
  return P;
}

// Expand `n` packed fingerprints (optionally gathered through `rows`) into workspace `ws`.
int prepare(const uint32_t* d_in, const int32_t* d_rows, int64_t n, int fpBits, void* ws, hipStream_t stream);

// Dense similarity on prepared sets.  metric = NVMK_METRIC_*.
int launch_dense(int metric, const Prepared& A, const Prepared& B, double* out, int64_t ld, hipStream_t stream);

// counts[id(x row)] += sign * #{ y rows that are neighbours } on prepared sets (matrix-core twin of
// butina.hip's neighbor_count_kernel).  Rows may be gathered through index lists (NULL = identity); the
// counts array is indexed by the PHYSICAL row id.  `symmetric`: x and y are the same un-gathered set, only
// tiles on or above the diagonal are evaluated and off-diagonal tiles credit both their rows and columns.
struct CountArgs {
  int             metric;
  float           thr;
  const uint16_t* table;    // Tanimoto threshold table with 4F + 3 entries (butina.hip), device
  const float*    tableF;   // the same thresholds as floats, +inf = never (what the kernel reads)
  int             sign;
  const int32_t*  xRows;    // logical -> physical row of X (NULL: identity)
  const int32_t*  xIds;     // logical row -> index into counts (NULL: the physical row)
  int64_t         nX;       // logical rows of x (host-side upper bound when nXdev is set)
  const int32_t*  nXdev;    // optional device-side row count
  const int32_t*  yRows;
  const int32_t*  yIds;     // used in symmetric mode only (column credits)
  int64_t         nY;
  const int32_t*  nYdev;
  bool            symmetric;
  // optional (symmetric mode only): also emit every neighbour pair i < j once as (i, j) into `edges`; `edgeCursor`
  // counts ALL pairs found (it may pass `edgeCapacity`: the caller checks and falls back), device pointers
  int2*               edges        = nullptr;
  unsigned long long* edgeCursor   = nullptr;
  unsigned long long  edgeCapacity = 0;
  // rows of X and Y are sorted by popcount (ascending): tiles whose popcount bands cannot reach the Tanimoto threshold
  // (T <= min(pa, pb) / max(pa, pb)) are skipped.  Tile count kernel, Tanimoto only.
  bool                bandSkip     = false;
  // symmetric mode only: evaluate just the tile rows [tileRowLo, tileRowHi) of the upper triangle (a row shard of the
  // all-pairs pass; 0, 0 = everything).  Row AND column credits of those tiles are added, so the per-shard counts of all
  // shards sum to the full degrees and their edge lists are disjoint.
  unsigned            tileRowLo    = 0;
  unsigned            tileRowHi    = 0;
};
int launch_counts(const CountArgs& args, const Prepared& X, const Prepared& Y, int32_t* counts, hipStream_t stream);

}  // namespace fp4
}  // namespace nvmk
