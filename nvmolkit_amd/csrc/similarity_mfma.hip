// Matrix-core (FP4 MFMA) formulation of fingerprint cross-similarity and neighbour counting — gfx950.
//
// See fp4.h for why popcount(a & b) maps exactly onto v_mfma_scale_f32_32x32x64_f8f6f4 and for the
// prepared-set layout.  Replaces the same reference kernels as similarity.hip
// (src/similarity_kernels.cu:96-240 is the reference's own tensor-core formulation) and the Triton
// neighbour-count kernel (nvmolkit/_fusedButina.py:99-179).
//
// Kernel shape: 256 threads = 4 wave64 in a 2 x 2 arrangement, workgroup tile 128 x 128, wave tile
// 64 x 64 = 2 x 2 MFMA blocks of 32 x 32 (4 accumulators of 16 VGPRs).  K is walked in LDS chunks of
// 8 words (256 fingerprint bits = 128 B of FP4 per row): per chunk each operand tile is 16 KB, one
// workgroup uses 33 KB so FOUR workgroups (16 waves) share a CU and cover each other's global->LDS latency
// and store drain.  A k-step (64 bits) is one MFMA per block: lanes 0-31 hold word 2t of rows 0-31, lanes
// 32-63 word 2t+1; which nibble carries which k is irrelevant because both operands use the same expansion.
// LDS rows are XOR-swizzled in 16-byte slots: every ds_read_b128 lane group touches 16 different slots of the
// 256-byte bank row.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <type_traits>

#include "fp4.h"
#include "options.h"
#include "tile_maps.h"

namespace nvmk {
namespace fp4 {

namespace {

constexpr int TM       = 128;
constexpr int TN       = 128;
constexpr int NT       = 256;
constexpr int SUPER = 64;  // supertile edge in tiles: 64 x 64 tiles = 8 MB + 8 MB of FP4 operands at 2048 bits

typedef int   v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t spread8(uint32_t x) {  // bit k of the low byte -> 0x2 in nibble k
  x = (x | (x << 12)) & 0x000F000Fu;
  x = (x | (x << 6)) & 0x03030303u;
  x = (x | (x << 3)) & 0x11111111u;
  return x << 1;
}

// 16 lanes per fingerprint: lane s expands words s, s+16, ... and the group reduces the popcount.
__global__ __launch_bounds__(NT) void prepare_kernel(const uint32_t* __restrict__ in, const int32_t* __restrict__ rows,
                                                     const int64_t n, const int64_t nPad, const int W, const int Wp,
                                                     int32_t* __restrict__ popc, uint4* __restrict__ out) {
  const int64_t gid = static_cast<int64_t>(blockIdx.x) * NT + threadIdx.x;
  const int64_t row = gid >> 4;
  const int     sub = static_cast<int>(gid & 15);
  if (row >= nPad) return;
  const bool      live = row < n;
  const int64_t   src  = live ? (rows ? static_cast<int64_t>(rows[row]) : row) : 0;
  const uint32_t* r    = in + src * W;
  int             cnt  = 0;
  for (int w = sub; w < Wp; w += 16) {
    const uint32_t word = (live && w < W) ? r[w] : 0u;
    cnt += __popc(word);
    uint4 e;
    e.x               = spread8(word & 0xffu);
    e.y               = spread8((word >> 8) & 0xffu);
    e.z               = spread8((word >> 16) & 0xffu);
    e.w               = spread8(word >> 24);
    out[row * Wp + w] = e;
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (sub == 0) popc[row] = cnt;
}

__device__ __forceinline__ v16f mfma_fp4(const uint4 a, const uint4 b, const v16f c) {
  const v8i av = {static_cast<int>(a.x), static_cast<int>(a.y), static_cast<int>(a.z), static_cast<int>(a.w), 0, 0, 0, 0};
  const v8i bv = {static_cast<int>(b.x), static_cast<int>(b.y), static_cast<int>(b.z), static_cast<int>(b.w), 0, 0, 0, 0};
  // cbsz = blgp = 4 selects FP4 e2m1 for A and B; E8M0 scale 0x7f = 2^0
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// Geometry of one LDS K-chunk of KCW words: KCW 16-byte slots per row, XOR-swizzled so that the 16 lanes
// of a ds_read_b128 lane group (16 rows with distinct row & 15) land on 16 different slots of the
// 256-byte bank row.  KCW = 16: one row per bank row, slot ^= row & 15.  KCW = 8: two rows per bank row
// (row & 1 picks the half), slot ^= (row >> 1) & 7.
template <int KCW> struct Chunk {
  static constexpr int ROWBYTES = KCW * 16;
  static constexpr int LOG      = (KCW == 16) ? 4 : (KCW == 8 ? 3 : 2);
  static constexpr int T        = TM * KCW / NT;  // 16-byte pieces per thread per operand
  __device__ static __forceinline__ unsigned swz(const unsigned row) {
    return (KCW == 16) ? (row & 15u) : (KCW == 8 ? ((row >> 1) & 7u) : ((row >> 2) & 3u));
  }
};

// One K chunk of MFMAs for this wave's 64 x 64 tile.
template <int KCW>
__device__ __forceinline__ void chunk_mma(v16f (&acc)[2][2], const char* sA, const char* sB, const int wm, const int wn,
                                          const int lane) {
  using C               = Chunk<KCW>;
  const int      l31    = lane & 31;
  const unsigned half   = static_cast<unsigned>(lane >> 5);
  const unsigned rowA0  = static_cast<unsigned>(wm * 64 + l31);
  const unsigned rowB0  = static_cast<unsigned>(wn * 64 + l31);
  // rows +32 keep the swizzle term, so one term serves both blocks of an operand
  const unsigned baseA  = rowA0 * C::ROWBYTES;
  const unsigned baseB  = rowB0 * C::ROWBYTES;
  const unsigned swA    = C::swz(rowA0);
  const unsigned swB    = C::swz(rowB0);
#pragma unroll
  for (int ks = 0; ks < KCW / 2; ++ks) {
    const unsigned slot = static_cast<unsigned>(ks * 2) + half;
    const unsigned offA = baseA + ((slot ^ swA) << 4);
    const unsigned offB = baseB + ((slot ^ swB) << 4);
    const uint4    a0   = *reinterpret_cast<const uint4*>(sA + offA);
    const uint4    a1   = *reinterpret_cast<const uint4*>(sA + offA + 32 * C::ROWBYTES);
    const uint4    b0   = *reinterpret_cast<const uint4*>(sB + offB);
    const uint4    b1   = *reinterpret_cast<const uint4*>(sB + offB + 32 * C::ROWBYTES);
    acc[0][0]           = mfma_fp4(a0, b0, acc[0][0]);
    acc[0][1]           = mfma_fp4(a0, b1, acc[0][1]);
    acc[1][0]           = mfma_fp4(a1, b0, acc[1][0]);
    acc[1][1]           = mfma_fp4(a1, b1, acc[1][1]);
  }
}

// ---- dense cross-similarity ----------------------------------------------------------------------
// Operand chunks go global -> LDS directly (global_load_lds_dwordx4: 1 KB per wave instruction, no staging
// VGPRs, no ds_write pass).  The DMA destination is wave-uniform base + lane * 16, so the LDS image is linear in
// "piece" order and the XOR swizzle is applied to the per-lane SOURCE slot instead (slot ^ swz(row) is an
// involution; chunk_mma applies the same one on the read side).  110 VGPRs and 33 KB of LDS -> 4 workgroups
// per CU.  Measured alternatives at 1M x 1M, 2048 bits (T pairs/s): VGPR-staged loads + ds_write with a
// reciprocal table in LDS, 3 workgroups/CU 0.54; this kernel 0.60; double-buffered 8-word chunks (65 KB, 2
// workgroups/CU) 0.53; double-buffered 4-word chunks 0.56; single-buffered 4-word chunks at 5 workgroups/CU
// 0.57; plain instead of nontemporal stores 0.47 (the output stream evicts the operand blocks from L2).
// A pure store kernel with this tile/lane pattern reaches 5.44 TB/s (tools/ubench_store.hip), this kernel 4.9.
//
// The Tanimoto ratio needs no table: r0 = v_rcp_f32(u) (1 ulp), one Newton step in f64 (relative error
// < 2^-44), then q0 = c r, e = fma(-q0, u, c), q = fma(e, r, q0): the value before the last rounding is
// within 2^-88 relative of c / u, and a quotient of integers < 2^24 is either exactly representable or
// > 2^-80 (relative) away from every rounding boundary, so q is the correctly rounded IEEE quotient.
// Checked exhaustively for u <= 16384 on the CPU for every possible 1-ulp seed (oracle_similarity.c
// orc_check_newton_division) and for u <= 4096 on the device
// (tests/test_similarity_gpu.py::test_prefix_fingerprints_exhaust_all_ratios).
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void*       lptr_t;

__device__ __forceinline__ double ratio_by_newton(const int c, const int u) {
  const double ud = static_cast<double>(u);
  const double cd = static_cast<double>(c);
  const double r0 = static_cast<double>(__builtin_amdgcn_rcpf(static_cast<float>(u)));
  const double r  = __fma_rn(__fma_rn(-ud, r0, 1.0), r0, r0);
  const double q0 = __dmul_rn(cd, r);
  const double e  = __fma_rn(-q0, ud, cd);
  return __fma_rn(e, r, q0);
}
// Same quotient from the f32 accumulator and an f32 popcount sum (both exact integers < 2^24): two conversions fewer
// per element.  (v_rcp_f64 instead of the f32 seed + Newton step was measured 2.7 % faster but is not accurate enough:
// 3/49 came out one ulp off.)
__device__ __forceinline__ double ratio_by_newton_f(const float c, const float u) {
  const double ud = static_cast<double>(u);
  const double cd = static_cast<double>(c);
  const double r0 = static_cast<double>(__builtin_amdgcn_rcpf(u));
  const double r  = __fma_rn(__fma_rn(-ud, r0, 1.0), r0, r0);
  const double q0 = __dmul_rn(cd, r);
  const double e  = __fma_rn(-q0, ud, cd);
  return __fma_rn(e, r, q0);
}

template <int METRIC>
__global__ __launch_bounds__(NT, 4) void cross_sim_mfma_kernel(const uint4* __restrict__ A, const int32_t* __restrict__ popA,
                                                               const int64_t nA, const uint4* __restrict__ B,
                                                               const int32_t* __restrict__ popB, const int64_t nB,
                                                               const int Wp, double* __restrict__ out, const int64_t ld,
                                                               const unsigned tilesM, const unsigned tilesN, const unsigned superM) {
  // One 8-word chunk buffer: load -> barrier -> multiply; the phases of the 4 co-resident workgroups overlap each other
  // (in-workgroup rings, wider tiles and a producer / consumer split were measured slower: tools/experiments/).
  constexpr int KCW = 8;
  constexpr int PPW = KCW / 2;   // DMA pieces (1 KB wave instructions) per wave and operand (32 rows per wave)
  constexpr int NWV = 4;         // waves: 2 x 2 of 64 x 64
  constexpr int BPW = PPW;
  constexpr int RPP = 64 / KCW;  // rows per piece
  using C           = Chunk<KCW>;
  constexpr int STAGE = (TM + TN) * C::ROWBYTES;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* sA  = smem;
  char* sB  = smem + TM * C::ROWBYTES;
  float* pcA = reinterpret_cast<float*>(smem + STAGE);  // popcounts as the f32 values the epilogue works with
  float* pcB = pcA + TM;

  // Workgroup -> tile map: blockIdx.y walks 64 x 64-tile supertiles (both operand blocks of a supertile, 8 MB each, stay in
  // L2 / Infinity Cache: a 1M-row operand streamed tile by tile made EVERY chunk load a first touch, 0.41 vs 0.58 T pairs/s).
  // Inside a supertile the map is XCD-aware: workgroup b runs on XCD b % 8 and every XCD has its own L2, so XCD x gets the
  // 32 x 16-tile sub-block x of the supertile (2 x 4 sub-blocks) and walks it with tile_n fastest.  Its L2 then holds the
  // sub-block's 16 B tiles (2 MB, reused 32 times) and the current A tile: (32 + 16) tile loads per XCD and supertile
  // instead of (64 + 8) with the XCDs interleaved over tile_n — a third less traffic from the L2s into the fabric.
  // With 128 or more tile rows in the launch (16 384-row chunks) the supertile is 128 tiles tall (superM) and an XCD owns
  // 64 x 16 tiles: its 16 B tiles are then reused 64 times, (64 + 16) tile loads per 1024 tiles instead of (32 + 16) per 512.
  unsigned tile_m, tile_n;
  maps::dense_tile(blockIdx.x, blockIdx.y, tilesN, superM, tile_m, tile_n);  // tile_maps.h (host-tested: tests/test_tile_maps.py)
  if (tile_m >= tilesM || tile_n >= tilesN) return;

  const int     tid   = threadIdx.x;
  const int     lane  = tid & 63;
  const int     wave  = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int     wm    = wave >> 1;
  const int     wn    = wave & 1;
  const int64_t rowA0 = static_cast<int64_t>(tile_m) * TM;
  const int64_t rowB0 = static_cast<int64_t>(tile_n) * TN;

  if (tid < TM) {
    // Tanimoto: an empty row gets 0.5 instead of 0, so that the union pa + pb - c is never 0 and the epilogue needs no
    // per-element clamp: c is 0 whenever a popcount is 0, and 0 / u is exactly 0 for every u > 0
    const float pa = static_cast<float>(popA[rowA0 + tid]);
    pcA[tid]       = METRIC == NVMK_METRIC_TANIMOTO ? fmaxf(pa, 0.5f) : pa;
  } else if (tid - TM < TN) {
    pcB[tid - TM] = static_cast<float>(popB[rowB0 + tid - TM]);
  }
  // Wave priority by phase: a workgroup in its main loop (DMA issue, ds_read, MFMA) goes before co-resident workgroups
  // that are converting and storing, which have plenty of independent work to hide behind (+0.8 % measured).
  __builtin_amdgcn_s_setprio(2);

  v16f acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
    }
  }

  {
    // piece p = (wave * PPW + t) * 64 + lane lands at LDS byte 16 p: row p / KCW, physical slot p % KCW;
    // the swizzle term differs per piece (row + RPP t) and is folded in per t
    const unsigned prow    = static_cast<unsigned>(wave * 32 + (lane >> C::LOG));
    const int64_t  rowStep = static_cast<int64_t>(RPP) * Wp;
    const uint4*   gA      = A + (rowA0 + prow) * Wp;
    const unsigned prowB   = static_cast<unsigned>(wave * (TN / NWV) + (lane >> C::LOG));
    // LDS row p of the B tile holds tile COLUMN (p & 64) + 2 (p & 31) + ((p >> 5) & 1): the two 32-column MFMA blocks of a
    // wave then cover the even and the odd columns of its 64, so a lane's accumulators acc[.][0][r], acc[.][1][r] are
    // ADJACENT columns of one row and leave as one 16-byte store — no lane exchange, no selects in the epilogue.
    const unsigned colB    = (prowB & 64u) + 2u * (prowB & 31u) + ((prowB >> 5) & 1u);
    const uint4*   gB      = B + (rowB0 + colB) * Wp;
    const int      nChunks = Wp / KCW;
    for (int ch = 0; ch < nChunks; ++ch) {
      if (ch > 0) __syncthreads();  // every wave is done reading the previous chunk
#pragma unroll
      for (int t = 0; t < PPW; ++t) {
        const unsigned slot = (static_cast<unsigned>(lane) & (KCW - 1u)) ^ C::swz(prow + RPP * t);
        __builtin_amdgcn_global_load_lds((gptr_t)(gA + t * rowStep + ch * KCW + slot),
                                         (lptr_t)(sA + (wave * PPW + t) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < BPW; ++t) {
        const unsigned slot = (static_cast<unsigned>(lane) & (KCW - 1u)) ^ C::swz(prowB + RPP * t);
        __builtin_amdgcn_global_load_lds((gptr_t)(gB + 2 * t * rowStep + ch * KCW + slot),  // RPP LDS rows = 2 RPP columns
                                         (lptr_t)(sB + (wave * BPW + t) * 1024), 16, 0, 0);
      }
      __syncthreads();  // hipcc drains vmcnt before the barrier: the chunk has landed for every wave
      chunk_mma<KCW>(acc, sA, sB, wm, wn, lane);
    }
  }

  // Epilogue: D[i][j], i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), j = lane & 31 inside each 32 x 32 block.
  // Addresses are a wave-uniform 64-bit row base plus one 32-bit per-lane byte offset; interior tiles take the
  // branch-free path so the 64 divisions and stores of a lane interleave.
  __builtin_amdgcn_s_setprio(0);
  const bool     full    = (rowA0 + TM <= nA) && (rowB0 + TN <= nB);
  const unsigned hi      = static_cast<unsigned>(lane >> 5);
  const int      col0    = 2 * (lane & 31);  // this lane's two columns inside the wave tile: col0 (block 0), col0 + 1 (block 1)
  const unsigned laneOff = (hi * 4u * static_cast<unsigned>(ld) + static_cast<unsigned>(col0)) * 8u;
  char*          waveOut = reinterpret_cast<char*>(out + (rowA0 + wm * 64) * ld + rowB0 + wn * 64);
  // popcounts and the f32 accumulators are exact integers < 2^24, so the union is formed in f32 (no int round trip)
  auto value = [&](const float c, const float pav, const float pbv) -> double {
    if constexpr (METRIC == NVMK_METRIC_TANIMOTO) {
      return ratio_by_newton_f(c, pav + pbv - c);  // u >= 0.5 (see pcA)
    } else {
      const double denom = sqrt(static_cast<double>(pav) * static_cast<double>(pbv));
      return (c == 0.0f || denom == 0.0) ? 0.0 : static_cast<double>(c) / denom;
    }
  };
  const float pb0 = pcB[wn * 64 + col0];
  const float pb1 = pcB[wn * 64 + col0 + 1];
  typedef double d2_t __attribute__((ext_vector_type(2)));
  // Interior tiles: 32 sixteen-byte stores per lane, each wave instruction writing two 512-byte row segments (rows i and
  // i + 4), branch-free so that the divisions and stores of a lane interleave.  Nontemporal: the 8 B/pair output stream
  // must not evict the operand blocks from L2.
  auto emit = [&](auto fullTag) {
    constexpr bool FULL = decltype(fullTag)::value;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int   il0    = mi * 32 + (r & 3) + 8 * (r >> 2);  // + 4 hi: this lane's row inside the wave tile
        const float pav    = pcA[wm * 64 + il0 + 4 * static_cast<int>(hi)];
        char*       rowOut = waveOut + static_cast<int64_t>(il0) * ld * 8;
        d2_t        v;
        v.x = value(acc[mi][0][r], pav, pb0);
        v.y = value(acc[mi][1][r], pav, pb1);
        if constexpr (FULL) {
          __builtin_nontemporal_store(v, reinterpret_cast<d2_t*>(rowOut + laneOff));
        } else if (rowA0 + wm * 64 + il0 + 4 * static_cast<int>(hi) < nA) {
          double* dst = reinterpret_cast<double*>(rowOut + laneOff);
          if (rowB0 + wn * 64 + col0 < nB) __builtin_nontemporal_store(v.x, dst);
          if (rowB0 + wn * 64 + col0 + 1 < nB) __builtin_nontemporal_store(v.y, dst + 1);
        }
      }
    }
  };
  if (full) {
    emit(std::true_type{});
  } else {
    emit(std::false_type{});
  }
}

// ---- neighbour counting on the matrix cores ------------------------------------------------------
// Same main loop as the dense kernel (LDS-DMA operand chunks; the per-lane source row goes through the optional
// gather lists, so the Butina loop never compacts the fingerprint matrix), but the epilogue thresholds the exact
// counts with the Tanimoto table tmin[pa + pb] (see butina.hip; read through L1, it is 16 KB) and reduces them to
// per-row (and, in symmetric mode, per-column) neighbour counts: ballot + scalar popcount -> LDS -> one global atomic
// per row.  34 KB of LDS -> 4 workgroups per CU.

__device__ __forceinline__ bool cosine_neighbor(const int c, const int pa, const int pb, const float thr) {
  const float denom = sqrtf(static_cast<float>(pa) * static_cast<float>(pb));
  if (!(denom > 0.0f)) return false;
  return static_cast<float>(c) / denom >= thr;
}

// Tanimoto predicate without a table (ARITH): float(c)/float(s - c) >= thr  <=>  c/(s - c) >= m (or > m), m = the
// rounding boundary just below thr  <=>  c (1 + m) - pb m > pa m - adj, with K1 = 1 + m and K2 = m exact doubles, every
// product and difference exact in f64 for s <= 2^13 and thr >= 2^-10, and adj = half a grid unit for ">=" / 0 for ">"
// (ArithThreshold below; checked exhaustively against real float divisions by oracle/orc_check_threshold_arith).
// Neighbours are rare, so a row slot (one accumulator row across the wave's column blocks) first takes the max of its
// lanes' values and leaves after ONE compare when no lane has a hit: the 64 table lookups through the texture path,
// ballots and scalar popcounts per lane and tile of the table form cost a third of the kernel.
struct ArithThreshold {
  double k1, k2, adj;
  bool   ok;  // false: outside the exact range, use the table
};
inline ArithThreshold arith_threshold(const float thr, const int F) {
  ArithThreshold t{0.0, 0.0, 0.0, false};
  if (!(thr >= 0x1p-10f && thr <= 1.0f) || F > 4096) return t;
  const float  pred = std::nextafterf(thr, -std::numeric_limits<float>::infinity());
  const double m    = 0.5 * (static_cast<double>(thr) + static_cast<double>(pred));  // exact
  uint32_t     bits;
  std::memcpy(&bits, &thr, sizeof(bits));
  const double grid = 0.5 * (static_cast<double>(thr) - static_cast<double>(pred));  // every quantity is a multiple of it
  t.k1  = 1.0 + m;
  t.k2  = m;
  t.adj = (bits & 1u) ? 0.0 : 0.5 * grid;  // even significand: the tie at m rounds up to thr, so ">= m" = "> m - grid/2"
  t.ok  = true;
  return t;
}

constexpr int EDGE_STAGE = 256;  // neighbour pairs a workgroup of the tile count kernel stages in LDS

template <int METRIC, bool EMIT, bool ARITH = false>
__global__ __launch_bounds__(NT, 4) void neighbor_count_mfma_kernel(
  const uint4* __restrict__ X, const int32_t* __restrict__ popX, const int32_t* __restrict__ xRows,
  const int32_t* __restrict__ xIds, int64_t nX,
  const int32_t* __restrict__ nXdev, const uint4* __restrict__ Y, const int32_t* __restrict__ popY,
  const int32_t* __restrict__ yRows, const int32_t* __restrict__ yIds, int64_t nY, const int32_t* __restrict__ nYdev, const int Wp, const int F,
  const float* __restrict__ table, const float thr, const int sign, const int symmetric, int32_t* __restrict__ counts, const unsigned superN,
  const unsigned superW, const unsigned superH, int2* __restrict__ edges, unsigned long long* __restrict__ edgeCursor,
  const unsigned long long edgeCapacity, const double K1, const double K2, const double adj, const float bandThr,
  const unsigned tileRowLo, const unsigned tileRowHi) {
  constexpr int KCW = 8;
  constexpr int PPW = KCW / 2;
  constexpr int RPP = 64 / KCW;
  using C           = Chunk<KCW>;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* sA     = smem;
  char* sB     = smem + TM * C::ROWBYTES;
  int*  pcA    = reinterpret_cast<int*>(smem + (TM + TN) * C::ROWBYTES);
  int*  pcB    = pcA + TM;
  int*  rowsum = pcB + TN;
  int*  colsum = rowsum + TM;
  // EMIT: neighbour pairs of the tile are staged here and leave with ONE atomic on the global cursor per workgroup
  // (one per non-empty ballot serialised 2.3 M returning atomics on a single address at N = 100k: 27 ms for a 5 ms pass)
  int2* edgeBuf  = reinterpret_cast<int2*>(colsum + TN);  // [EDGE_STAGE]
  int*  edgeMeta = reinterpret_cast<int*>(edgeBuf + EDGE_STAGE);  // [0] staged, [1..2] global base

  if (nXdev) nX = *nXdev;
  if (nYdev) nY = *nYdev;
  const unsigned tilesM = static_cast<unsigned>((nX + TM - 1) / TM);
  const unsigned tilesN = static_cast<unsigned>((nY + TN - 1) / TN);
  // supertile map over the HOST-side upper bounds (gridDim), exits against the device-side sizes.  A
  // supertile is superH x superW tiles; superW < superH for skinny problems (a Butina subtract pass has one
  // column tile: a square 64 x 64 map would launch 63 empty workgroups per working one).
  // One workgroup per tile.  (A persistent variant — 4 workgroups per CU striding over the tile list — was measured
  // 40 % SLOWER: statically strided workgroups run phase-locked, so the four on a CU load together and then compute
  // together instead of covering each other.)
  const unsigned sidx   = blockIdx.z * gridDim.y + blockIdx.y;  // supertile index (grid y and z are 16-bit each)
  unsigned       sm, sn;
  if (symmetric) {
    // only the supertiles on or above the diagonal are launched (the strictly lower ones would be 2 M workgroups that
    // exit at once at N = 1M): sidx enumerates row sm = 0.., columns sn = sm..superN-1
    if (!maps::symmetric_supertile(sidx, superN, sm, sn)) return;  // padding of the 2-D supertile grid (tile_maps.h)
  } else {
    sm = sidx / superN;
    sn = sidx - sm * superN;
  }
  unsigned tile_m, tile_n;
  // XCD-aware walk of a full supertile, as in the dense kernel: workgroup b runs on XCD b % 8, which gets its own
  // 32 x 16-tile sub-block, so that its L2 keeps 16 column tiles and one row tile instead of sharing all 64 row tiles
  maps::count_tile(blockIdx.x, sm, sn, superH, superW, tile_m, tile_n);
  if (tile_m >= tilesM || tile_n >= tilesN) return;
  if (symmetric && tile_n < tile_m) return;
  if (tileRowHi != 0u && (tile_m < tileRowLo || tile_m >= tileRowHi)) return;  // another shard's tile rows
  const bool creditCols = symmetric && tile_n > tile_m;

  const int     tid   = threadIdx.x;
  const int     lane  = tid & 63;
  const int     wave  = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int     wm    = wave >> 1;
  const int     wn    = wave & 1;
  const int64_t rowA0 = static_cast<int64_t>(tile_m) * TM;
  const int64_t rowB0 = static_cast<int64_t>(tile_n) * TN;
  const int     SENT  = (METRIC == NVMK_METRIC_TANIMOTO) ? 2 * F + 1 : 0;  // popcount of a padded row

  // physical row of logical row r (rows past the end alias the last one; their popcount is the sentinel)
  auto physX = [&](const int64_t r) -> int {
    const int64_t rc = r < nX ? r : nX - 1;
    return xRows ? xRows[rc] : static_cast<int>(rc);
  };
  auto physY = [&](const int64_t r) -> int {
    const int64_t rc = r < nY ? r : nY - 1;
    return yRows ? yRows[rc] : static_cast<int>(rc);
  };
  if (tid < TM) {
    const int64_t r = rowA0 + tid;
    pcA[tid]        = r < nX ? popX[physX(r)] : SENT;
    rowsum[tid]     = 0;
  } else {
    const int     t = tid - TM;
    const int64_t r = rowB0 + t;
    pcB[t]          = r < nY ? popY[physY(r)] : SENT;
    colsum[t]       = 0;
  }
  if (EMIT && tid == 0) edgeMeta[0] = 0;
  if (bandThr > 0.0f) {
    // Rows sorted by popcount: a tile's popcounts span [pc[0], pc[last valid]].  Tanimoto <= min(pa, pb) / max(pa, pb),
    // so when the bands are further apart than the threshold (taken 1e-6 low: the f32 predicate cannot round across
    // that) no pair of the tile is a neighbour and the workgroup leaves before touching the operands.
    __syncthreads();
    const int    lastA = static_cast<int>(nX - rowA0 < TM ? nX - rowA0 : TM) - 1, lastB = static_cast<int>(nY - rowB0 < TN ? nY - rowB0 : TN) - 1;
    const double minA = pcA[0], maxA = pcA[lastA], minB = pcB[0], maxB = pcB[lastB], t = static_cast<double>(bandThr) * (1.0 - 1.0e-6);
    if (maxB < t * minA || maxA < t * minB) return;
  }

  v16f acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
    }
  }

  {
    // piece p = (wave * PPW + t) * 64 + lane lands at LDS byte 16 p: row p / KCW, physical slot p % KCW; its source
    // is the (gathered) row's logical slot physical ^ swz(row)
    const unsigned prow = static_cast<unsigned>(wave * 32 + (lane >> C::LOG));
    unsigned       oa[PPW], ob[PPW];  // offsets in uint4 units: < 2^31 rows * Wp is checked by the launcher
#pragma unroll
    for (int t = 0; t < PPW; ++t) {
      const unsigned row  = prow + RPP * t;
      const unsigned slot = (static_cast<unsigned>(lane) & (KCW - 1u)) ^ C::swz(row);
      oa[t]               = static_cast<unsigned>(physX(rowA0 + row)) * static_cast<unsigned>(Wp) + slot;
      ob[t]               = static_cast<unsigned>(physY(rowB0 + row)) * static_cast<unsigned>(Wp) + slot;
    }
    const int nChunks = Wp / KCW;
    for (int ch = 0; ch < nChunks; ++ch) {
      if (ch > 0) __syncthreads();
#pragma unroll
      for (int t = 0; t < PPW; ++t) {
        __builtin_amdgcn_global_load_lds((gptr_t)(X + oa[t] + ch * KCW), (lptr_t)(sA + (wave * PPW + t) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < PPW; ++t) {
        __builtin_amdgcn_global_load_lds((gptr_t)(Y + ob[t] + ch * KCW), (lptr_t)(sB + (wave * PPW + t) * 1024), 16, 0, 0);
      }
      __syncthreads();
      chunk_mma<KCW>(acc, sA, sB, wm, wn, lane);
    }
  }

  // Row counts without LDS traffic: a ballot of the predicate holds one row per 32-lane half (lanes 0-31: row il,
  // lanes 32-63: row il + 4), its two popcounts are scalar, and a select drops them into the lane that owns the row.
  // (Shuffle reductions cost 160 ds_bpermute per lane and tile: the 1M x 1M pass ran at 13 us per tile and CU against
  // 5 us for the dense kernel with the same main loop.)
  int       cc[2] = {0, 0};
  int       myRow = 0;  // lane L: neighbours of row wm * 64 + L found in this tile
  const int pb0   = pcB[wn * 64 + (lane & 31)];
  const int pb1   = pcB[wn * 64 + 32 + (lane & 31)];
  // The epilogue is VALU-issue bound (4 waves per SIMD run it back to back: 40 VALU cycles per pair cost 3.4 us per
  // tile and CU).  Kept lean: thresholds are floats compared against the f32 accumulators (exact integers, no
  // conversion), the table offset is one add per pair (byte offsets of row and column popcounts prepared once).
  const unsigned pbOff0 = static_cast<unsigned>(pb0) * 4u, pbOff1 = static_cast<unsigned>(pb1) * 4u;
  const char*    tabB   = reinterpret_cast<const char*>(table);
  // ARITH: popcounts that can have no neighbour (padding sentinel, empty fingerprint) become +inf
  const double pbK0 = (pb0 == 0 || pb0 >= SENT) ? __builtin_inf() : static_cast<double>(pb0) * K2;
  const double pbK1 = (pb1 == 0 || pb1 >= SENT) ? __builtin_inf() : static_cast<double>(pb1) * K2;
  // ... and a single-precision screen in front of the exact test: the accumulators are floats already, and with
  // |c (1 + m)|, |s m| < 2^13 the f32 evaluation of c (1 + m) - pb m - pa m is within 0.01 of the exact value, so a
  // slot whose best lane is below -0.02 holds no neighbour (2 FMAs + max + compare per slot instead of the f64 chain)
  const float K1f = static_cast<float>(K1), K2f = static_cast<float>(K2);
  const float pbF0 = (pb0 == 0 || pb0 >= SENT) ? __builtin_inff() : static_cast<float>(pb0) * K2f;
  const float pbF1 = (pb1 == 0 || pb1 >= SENT) ? __builtin_inff() : static_cast<float>(pb1) * K2f;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int      rowLo = mi * 32 + (r & 3) + 8 * (r >> 2);  // wave-local row of lanes 0-31; lanes 32-63 hold rowLo + 4
      const int      pav   = pcA[wm * 64 + rowLo + 4 * (lane >> 5)];
      const unsigned paOff = static_cast<unsigned>(pav) * 4u;
      double         d0 = 0.0, d1 = 0.0, paK = 0.0;
      if constexpr (ARITH) {
        const float paF = (pav == 0 || pav >= SENT) ? __builtin_inff() : static_cast<float>(pav) * K2f;
        const float s0 = fmaf(acc[mi][0][r], K1f, -pbF0), s1 = fmaf(acc[mi][1][r], K1f, -pbF1);
        if (__ballot(fmaxf(s0, s1) - paF > -0.02f) == 0) continue;  // screened out (the common case)
        paK = (pav == 0 || pav >= SENT) ? __builtin_inf() : __builtin_fma(static_cast<double>(pav), K2, -adj);
        d0  = __builtin_fma(static_cast<double>(acc[mi][0][r]), K1, -pbK0);
        d1  = __builtin_fma(static_cast<double>(acc[mi][1][r]), K1, -pbK1);
        if (__ballot(fmax(d0, d1) > paK) == 0) continue;  // no neighbour in this row slot (the common case)
      }
      int lo = 0, hi = 0;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        bool p;
        if constexpr (ARITH) {
          p = (ni ? d1 : d0) > paK;
        } else if constexpr (METRIC == NVMK_METRIC_TANIMOTO) {
          p = acc[mi][ni][r] >= *reinterpret_cast<const float*>(tabB + (paOff + (ni ? pbOff1 : pbOff0)));
        } else {
          p = cosine_neighbor(static_cast<int>(acc[mi][ni][r]), pav, ni ? pb1 : pb0, thr);
        }
        cc[ni] += p ? 1 : 0;
        const uint64_t m = __ballot(p);
        lo += __popc(static_cast<unsigned>(m));
        hi += __popc(static_cast<unsigned>(m >> 32));
        if constexpr (EMIT) {
          // neighbour pairs are rare (mean degree / N of all pairs): almost every ballot is empty and skips this.
          // Symmetric un-gathered mode only: logical row == physical row.  Pairs i < j once; the diagonal tile
          // holds both orientations and the self pairs, which are dropped here.
          if (m != 0) {
            const int64_t  gi = rowA0 + wm * 64 + rowLo + 4 * (lane >> 5);
            const int64_t  gj = rowB0 + wn * 64 + ni * 32 + (lane & 31);
            const bool     e  = p && gi < gj && gi < nX && gj < nY;
            const uint64_t me = __ballot(e);
            if (me != 0) {
              const int first = __ffsll(static_cast<long long>(me)) - 1;
              int       base  = 0;
              if (lane == first) base = atomicAdd(&edgeMeta[0], __popcll(me));  // LDS
              base = __shfl(base, first);
              if (e) {
                const int slot = base + __popcll(me & ((1ull << lane) - 1ull));
                if (slot < EDGE_STAGE) {
                  edgeBuf[slot] = make_int2(static_cast<int>(gi), static_cast<int>(gj));
                } else {  // a tile with more pairs than the staging area: straight to memory
                  const unsigned long long gs = atomicAdd(edgeCursor, 1ull);
                  if (gs < edgeCapacity) edges[gs] = make_int2(static_cast<int>(gi), static_cast<int>(gj));
                }
              }
            }
          }
        }
      }
      // drop the two scalar row counts into the lanes that own rows rowLo and rowLo + 4 (v_writelane: lane select in M0)
      asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(myRow) : "s"(lo), "s"(rowLo));
      asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(myRow) : "s"(hi), "s"(rowLo + 4));
    }
  }
  if (myRow != 0) atomicAdd(&rowsum[wm * 64 + lane], myRow);
  if (creditCols) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      int v = cc[ni];
      v += __shfl_xor(v, 32);
      if (lane < 32 && v != 0) atomicAdd(&colsum[wn * 64 + ni * 32 + lane], v);
    }
  }
  __syncthreads();
  if constexpr (EMIT) {
    const int staged = edgeMeta[0] < EDGE_STAGE ? edgeMeta[0] : EDGE_STAGE;
    if (staged > 0) {  // workgroup-uniform
      if (tid == 0) {
        const unsigned long long gb = atomicAdd(edgeCursor, static_cast<unsigned long long>(staged));
        edgeMeta[1] = static_cast<int>(gb & 0xffffffffull);
        edgeMeta[2] = static_cast<int>(gb >> 32);
      }
      __syncthreads();
      const unsigned long long gb = (static_cast<unsigned long long>(static_cast<unsigned>(edgeMeta[2])) << 32) | static_cast<unsigned>(edgeMeta[1]);
      for (int i = tid; i < staged; i += NT) {
        if (gb + i < edgeCapacity) edges[gb + i] = edgeBuf[i];
      }
    }
  }
  if (tid < TM) {
    const int     v = rowsum[tid];
    const int64_t r = rowA0 + tid;
    if (v != 0 && r < nX) atomicAdd(&counts[xIds ? xIds[r] : physX(r)], sign * v);
  } else if (creditCols) {
    const int     t = tid - TM;
    const int     v = colsum[t];
    const int64_t r = rowB0 + t;
    if (v != 0 && r < nY) atomicAdd(&counts[yIds ? yIds[r] : physY(r)], sign * v);
  }
}

#include "count_panel.inc"

}  // namespace

int prepare(const uint32_t* d_in, const int32_t* d_rows, int64_t n, int fpBits, void* ws, hipStream_t stream) {
  NVMK_REQUIRE(fpBits > 0 && fpBits % 32 == 0, "fp_bits must be a positive multiple of 32, got %d", fpBits);
  NVMK_REQUIRE(n >= 0, "negative row count");
  if (n == 0) return NVMK_OK;
  NVMK_REQUIRE(d_in != nullptr && ws != nullptr, "fp4 prepare: NULL buffer");
  const Layout  L      = layout(n, fpBits);
  auto*         popc   = static_cast<int32_t*>(ws);
  auto*         rows   = reinterpret_cast<uint4*>(static_cast<char*>(ws) + L.rowsOffset);
  const int64_t blocks = ceil_div<int64_t>(L.nPad * 16, NT);
  NVMK_REQUIRE(blocks <= 0x7fffffffLL, "fp4 prepare: too many rows (%lld)", (long long)n);
  hipLaunchKernelGGL(prepare_kernel, dim3(static_cast<unsigned>(blocks)), dim3(NT), 0, stream, d_in, d_rows, n, L.nPad,
                     L.W, L.Wp, popc, rows);
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

template <int METRIC>
int launch_dense_t(const Prepared& A, const Prepared& B, double* out, int64_t ld, dim3 grid, unsigned tilesM,
                   unsigned tilesN, unsigned superM, hipStream_t stream) {
  const size_t shmem = static_cast<size_t>(TM + TN) * 8 * 16 + (TM + TN) * 4;
  hipLaunchKernelGGL(cross_sim_mfma_kernel<METRIC>, grid, dim3(NT), shmem, stream, A.rows, A.popc, A.L.n, B.rows, B.popc,
                     B.L.n, A.L.Wp, out, ld, tilesM, tilesN, superM);
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

int launch_dense(int metric, const Prepared& A, const Prepared& B, double* out, int64_t ld, hipStream_t stream) {
  if (A.L.n == 0 || B.L.n == 0) return NVMK_OK;
  NVMK_REQUIRE(A.L.Wp == B.L.Wp && A.L.W == B.L.W, "prepared sets have different fingerprint widths");
  NVMK_REQUIRE(out != nullptr && ld >= B.L.n, "cross similarity: bad output buffer / ld_out");
  // the epilogue keeps 4 rows * ld * 8 bytes in a 32-bit lane offset; popcounts must stay exact in f32
  NVMK_REQUIRE(ld < (int64_t{1} << 26), "cross similarity: ld_out %lld too large (max 2^26 - 1)", (long long)ld);
  NVMK_REQUIRE(A.L.W * 32 < (1 << 24), "cross similarity: fingerprints too wide for the matrix-core path");
  const int64_t tilesM = ceil_div<int64_t>(A.L.n, TM);
  const int64_t tilesN = ceil_div<int64_t>(B.L.n, TN);
  const int64_t superM = maps::dense_super_m(tilesM);  // supertile height in tiles (tile_maps.h)
  const int64_t supers = ceil_div<int64_t>(tilesM, superM) * ceil_div<int64_t>(tilesN, SUPER);
  NVMK_REQUIRE(supers <= 65535, "cross similarity: problem too large for one launch (%lld x %lld tiles)",
               (long long)tilesM, (long long)tilesN);
  const dim3     grid(static_cast<unsigned>(superM * SUPER), static_cast<unsigned>(supers));
  const unsigned tm = static_cast<unsigned>(tilesM), tn = static_cast<unsigned>(tilesN), sM = static_cast<unsigned>(superM);
  if (metric == NVMK_METRIC_TANIMOTO) return launch_dense_t<NVMK_METRIC_TANIMOTO>(A, B, out, ld, grid, tm, tn, sM, stream);
  return launch_dense_t<NVMK_METRIC_COSINE>(A, B, out, ld, grid, tm, tn, sM, stream);
}

int launch_counts(const CountArgs& a, const Prepared& X, const Prepared& Y, int32_t* counts, hipStream_t stream) {
  if (a.nX <= 0 || a.nY <= 0) return NVMK_OK;
  NVMK_REQUIRE(X.L.Wp == Y.L.Wp && X.L.W == Y.L.W, "prepared sets have different fingerprint widths");
  NVMK_REQUIRE(counts != nullptr, "neighbor counts: NULL counts");
  NVMK_REQUIRE(a.metric != NVMK_METRIC_TANIMOTO || a.tableF != nullptr, "neighbor counts: missing threshold table");
  const int     F       = X.L.W * 32;
  const bool    emit    = a.edges != nullptr;
  NVMK_REQUIRE(!emit || (a.symmetric && a.xRows == nullptr && a.yRows == nullptr && a.edgeCursor != nullptr),
               "neighbor counts: edge emission needs the symmetric, un-gathered mode and a cursor");
  const int64_t tilesM  = ceil_div<int64_t>(a.nX, TM);
  const int64_t tilesN  = ceil_div<int64_t>(a.nY, TN);
  const int64_t superE = [] {  // supertile edge for this kernel (NVMK_COUNT_SUPER overrides for experiments)
    const long v = opt::get(opt::kCountSuper).num(0);
    return static_cast<int64_t>(v > 0 && v <= 256 ? v : SUPER);
  }();
  const int64_t superW  = std::min<int64_t>(tilesN, superE);
  const int64_t superM  = ceil_div<int64_t>(tilesM, superE);
  const int64_t superN  = ceil_div<int64_t>(tilesN, superW);
  const int64_t supers = a.symmetric ? superN * (superN + 1) / 2 : superM * superN;  // symmetric: upper triangle only
  const int64_t gy     = std::min<int64_t>(supers, 65535);
  const int64_t gz     = ceil_div<int64_t>(supers, gy);
  NVMK_REQUIRE(gz <= 65535, "neighbor counts: problem too large for one launch");
  const dim3   grid(static_cast<unsigned>(superE * superW), static_cast<unsigned>(gy), static_cast<unsigned>(gz));
  const size_t shmem = static_cast<size_t>(TM + TN) * 8 * 16 + 4 * 128 * 4 + (emit ? EDGE_STAGE * 8 + 16 : 0);
  NVMK_REQUIRE(std::max(X.L.nPad, Y.L.nPad) * X.L.Wp < (int64_t{1} << 32),
               "neighbor counts: prepared set too large for 32-bit piece offsets");
  using Kern = void (*)(const uint4*, const int32_t*, const int32_t*, const int32_t*, int64_t, const int32_t*, const uint4*,
                        const int32_t*, const int32_t*, const int32_t*, int64_t, const int32_t*, int, int, const float*,
                        float, int, int, int32_t*, unsigned, unsigned, unsigned, int2*, unsigned long long*,
                        unsigned long long, double, double, double, float, unsigned, unsigned);
  Kern kern;
  ArithThreshold at = arith_threshold(a.thr, F);
  if (opt::get(opt::kCountThreshold).is("table")) at.ok = false;  // tests: force the table form
  {
    // Row-panel form (count_panel.inc) for the all-pairs pass of one un-gathered set: NVMK_COUNT_KERNEL=panel forces it wherever
    // it applies, tile never uses it, unset = by size.
    const opt::Text which   = opt::get(opt::kCountKernel);
    const bool      sameSet = X.rows == Y.rows && X.popc == Y.popc && a.nX == a.nY;
    const bool      whole   = a.tileRowHi == 0u || a.tileRowHi >= static_cast<unsigned>(tilesM);
    const bool      applies = a.symmetric && sameSet && !a.xRows && !a.yRows && !a.xIds && !a.yIds && !a.nXdev && !a.nYdev &&
                         a.metric == NVMK_METRIC_TANIMOTO && at.ok && (X.L.Wp == 64 || X.L.Wp == 32 || X.L.Wp == 16) &&
                         a.tileRowLo % 2u == 0u && (whole || a.tileRowHi % 2u == 0u);
    // auto: the whole pass of a set with at least 1024 panels (four per CU: below that the sweeps are too few to fill the chip and the
    // 128 x 128 tiles win, 23.5 against 24.4 ms at 100 000 rows); a row shard of the pass stays on the tile kernel unless forced
    const bool large = whole && a.tileRowLo == 0u && a.nX >= 1024 * static_cast<int64_t>(panel::ROWS);
    if (applies && (which.is("panel") || (large && !which.is("tile")))) {
      static std::atomic<int> cus[64] = {};  // per device
      int                     dev = 0;
      NVMK_HIP_CHECK(hipGetDevice(&dev));
      int nCu = (dev >= 0 && dev < 64) ? cus[dev].load() : 0;
      if (nCu <= 0) {
        NVMK_HIP_CHECK(hipDeviceGetAttribute(&nCu, hipDeviceAttributeMultiprocessorCount, dev));
        if (nCu < 8) nCu = 8;
        if (dev >= 0 && dev < 64) cus[dev].store(nCu);
      }
      const int      slots   = nCu / 8;
      const unsigned panels  = static_cast<unsigned>(ceil_div<int64_t>(a.nX, panel::ROWS));
      const unsigned panelLo = a.tileRowLo / 2u, panelHi = whole ? panels : a.tileRowHi / 2u;
      using PKern = void (*)(const uint4*, const int32_t*, int64_t, int, int32_t*, int2*, unsigned long long*, unsigned long long, double,
                             double, double, float, unsigned, unsigned, int, int*);
      PKern pk;
      if (X.L.Wp == 64) {
        pk = emit ? neighbor_count_panel_kernel<true, 32> : neighbor_count_panel_kernel<false, 32>;
      } else if (X.L.Wp == 32) {
        pk = emit ? neighbor_count_panel_kernel<true, 16> : neighbor_count_panel_kernel<false, 16>;
      } else {
        pk = emit ? neighbor_count_panel_kernel<true, 8> : neighbor_count_panel_kernel<false, 8>;
      }
      NVMK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pk), hipFuncAttributeMaxDynamicSharedMemorySize, panel::LDS_BYTES));
      int* groupBarrier = nullptr;  // one arrival counter per XCD group, stream-ordered like the launch itself
      NVMK_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&groupBarrier), 16 * sizeof(int), stream));  // [8 + xcd]: that group gave up waiting
      NVMK_HIP_CHECK(hipMemsetAsync(groupBarrier, 0, 16 * sizeof(int), stream));
      hipLaunchKernelGGL(pk, dim3(static_cast<unsigned>(slots * 8)), dim3(256), panel::LDS_BYTES, stream, X.rows, X.popc, a.nX, a.sign, counts,
                         a.edges, a.edgeCursor, a.edgeCapacity, at.k1, at.k2, at.adj,
                         (a.bandSkip && a.thr > 0.0f) ? a.thr : 0.0f, panelLo, panelHi, slots, groupBarrier);
      const hipError_t launched = hipGetLastError();
      NVMK_HIP_CHECK(hipFreeAsync(groupBarrier, stream));
      NVMK_REQUIRE(launched == hipSuccess, "neighbor counts (panel kernel): %s", hipGetErrorString(launched));
      return NVMK_OK;
    }
  }
  if (a.metric == NVMK_METRIC_TANIMOTO && at.ok) {
    kern = emit ? neighbor_count_mfma_kernel<NVMK_METRIC_TANIMOTO, true, true> : neighbor_count_mfma_kernel<NVMK_METRIC_TANIMOTO, false, true>;
  } else if (a.metric == NVMK_METRIC_TANIMOTO) {
    kern = emit ? neighbor_count_mfma_kernel<NVMK_METRIC_TANIMOTO, true> : neighbor_count_mfma_kernel<NVMK_METRIC_TANIMOTO, false>;
  } else {
    kern = emit ? neighbor_count_mfma_kernel<NVMK_METRIC_COSINE, true> : neighbor_count_mfma_kernel<NVMK_METRIC_COSINE, false>;
  }
  hipLaunchKernelGGL(kern, grid, dim3(NT), shmem, stream, X.rows, X.popc, a.xRows, a.xIds, a.nX, a.nXdev, Y.rows, Y.popc,
                     a.yRows, a.yIds, a.nY, a.nYdev, X.L.Wp, F, a.tableF, a.thr, a.sign, a.symmetric ? 1 : 0, counts,
                     static_cast<unsigned>(superN), static_cast<unsigned>(superW), static_cast<unsigned>(superE), a.edges,
                     a.edgeCursor, a.edgeCapacity, at.k1, at.k2, at.adj,
                     (a.bandSkip && a.metric == NVMK_METRIC_TANIMOTO && a.thr > 0.0f && a.xRows == nullptr && a.yRows == nullptr) ? a.thr : 0.0f,
                     a.symmetric ? a.tileRowLo : 0u, a.symmetric ? a.tileRowHi : 0u);
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

}  // namespace fp4
}  // namespace nvmk

extern "C" {

size_t nvmk_fp4_workspace_bytes(int64_t n, int fp_bits) {
  if (n < 0 || fp_bits <= 0 || fp_bits % 32 != 0) return 0;
  return nvmk::fp4::layout(n, fp_bits).bytes;
}

int nvmk_fp4_prepare(const uint32_t* d_in, int64_t n, int fp_bits, void* d_workspace, void* stream) {
  NVMK_MARK_ENTRY();
  return nvmk::fp4::prepare(d_in, nullptr, n, fp_bits, d_workspace, nvmk::as_stream(stream));
}

int nvmk_cross_similarity_prepared_f64(int metric, const void* d_ws_a, int64_t nA_total, int64_t a_row0, int64_t a_rows,
                                       const void* d_ws_b, int64_t nB, int fp_bits, double* d_out, int64_t ld_out,
                                       void* stream) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(metric == NVMK_METRIC_TANIMOTO || metric == NVMK_METRIC_COSINE, "unknown metric %d", metric);
  NVMK_REQUIRE(fp_bits > 0 && fp_bits % 32 == 0, "fp_bits must be a positive multiple of 32, got %d", fp_bits);
  NVMK_REQUIRE(nA_total >= 0 && nB >= 0 && a_row0 >= 0 && a_rows >= 0 && a_row0 + a_rows <= nA_total,
               "bad row range [%lld, +%lld) of %lld", (long long)a_row0, (long long)a_rows, (long long)nA_total);
  NVMK_REQUIRE(a_row0 % nvmk::fp4::ROW_PAD == 0, "a_row0 must be a multiple of %d", nvmk::fp4::ROW_PAD);
  if (a_rows == 0 || nB == 0) return NVMK_OK;
  NVMK_REQUIRE(d_ws_a && d_ws_b && d_out, "NULL buffer");
  nvmk::fp4::Prepared A = nvmk::fp4::view(d_ws_a, nA_total, fp_bits);
  A.popc += a_row0;
  A.rows += a_row0 * A.L.Wp;
  A.L.n    = a_rows;
  A.L.nPad = (a_rows + nvmk::fp4::ROW_PAD - 1) / nvmk::fp4::ROW_PAD * nvmk::fp4::ROW_PAD;
  return nvmk::fp4::launch_dense(metric, A, nvmk::fp4::view(d_ws_b, nB, fp_bits), d_out, ld_out, nvmk::as_stream(stream));
}

}  // extern "C"
