// EXPERIMENTS — not part of libnvmolkit_amd.so.
//
// The measured-and-rejected variants of the FP4 matrix-core kernels, kept buildable as lab notes:
//   * cross_sim_pp_kernel            persistent producer / consumer dense kernel (NVMK_DENSE_KERNEL=pp)      0.38-0.49 T pairs/s
//   * cross_sim_mfma_kernel<.,3>     three-stage in-workgroup ring of 4-word chunks (NVMK_DENSE_KERNEL=pipe) 0.56
//   * cross_sim_mfma_kernel<.,0,256> 256 x 128 tiles, 8 waves, 2 workgroups per CU (NVMK_DENSE_KERNEL=wide) 0.58
//   * neighbor_count_ring_kernel     pipelined 256 x 256 count kernel (NVMK_COUNT_KERNEL=ring, NVMK_RING_WT=64) 0.93 / 0.81
// against 0.64 (dense) and 1.15 (count) for the 128 x 128 tile kernels that ship.  This file is the round-1
// similarity_mfma.hip verbatim with its C entry points renamed nvmkx_*; tools/experiments/build.sh links it with the product's
// runtime.cpp into tools/experiments/libnvmk_experiments.so, tools/experiments/run_dense_variants.py times the variants.
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
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <type_traits>

#include "../../nvmolkit_amd/csrc/fp4.h"

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

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vmcnt_dyn(const int n) {  // n <= 32: whatever the ring geometry can produce
#define NVMK_W(k) case k: wait_vmcnt<k>(); break;
  switch (n) {
    NVMK_W(0) NVMK_W(1) NVMK_W(2) NVMK_W(3) NVMK_W(4) NVMK_W(5) NVMK_W(6) NVMK_W(7) NVMK_W(8) NVMK_W(9) NVMK_W(10)
    NVMK_W(11) NVMK_W(12) NVMK_W(13) NVMK_W(14) NVMK_W(15) NVMK_W(16) NVMK_W(17) NVMK_W(18) NVMK_W(19) NVMK_W(20)
    NVMK_W(21) NVMK_W(22) NVMK_W(23) NVMK_W(24) NVMK_W(25) NVMK_W(26) NVMK_W(27) NVMK_W(28) NVMK_W(29) NVMK_W(30)
    NVMK_W(31) NVMK_W(32)
    default: wait_vmcnt<0>(); break;
  }
#undef NVMK_W
}

// LDS-DMA issued through inline assembly.  With the builtin, the compiler's waitcnt pass tracks the DMA as a pending
// LDS write and puts s_waitcnt vmcnt(..) in front of every later ds_read of the same wave that it cannot prove
// disjoint — i.e. it drains the loads just issued for a LATER stage before the MFMAs of the current one, which
// serialises exactly what a ring is there to overlap.  The asm form is opaque to that pass; completion is handled by
// the counted waits below.  ldsByte = wave-uniform LDS byte address of the 1 KB (64 lanes x 16 B) destination.
__device__ __forceinline__ void dma_b128(const void* g, const unsigned ldsByte) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(ldsByte) : "memory");
}
__device__ __forceinline__ void dma_b32(const void* g, const unsigned ldsByte) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(ldsByte) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p)));
}

// workgroup barrier that orders LDS traffic only: __syncthreads() would also drain vmcnt, i.e. the DMA loads in flight
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

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

template <int METRIC, int PIPE = 0, int TMV = TM>
__global__ __launch_bounds__(NT * (TMV / TM), PIPE ? 3 : (TMV == TM ? 4 : 2)) void cross_sim_mfma_kernel(const uint4* __restrict__ A, const int32_t* __restrict__ popA,
                                                               const int64_t nA, const uint4* __restrict__ B,
                                                               const int32_t* __restrict__ popB, const int64_t nB,
                                                               const int Wp, double* __restrict__ out, const int64_t ld,
                                                               const unsigned tilesM, const unsigned tilesN) {
  // PIPE = 0: one 8-word chunk buffer, load -> barrier -> multiply (overlap only across the 4 co-resident workgroups).
  // PIPE = 3: ring of three 4-word chunks, two in flight while the third is multiplied (3 workgroups per CU).
  constexpr int KCW = PIPE ? 4 : 8;
  constexpr int PPW = KCW / 2;   // DMA pieces (1 KB wave instructions) per wave of the A operand (32 rows per wave)
  constexpr int NWV = 4 * (TMV / TM);  // waves: (TMV / 64) x 2 of 64 x 64
  constexpr int BPW = PPW * 4 / NWV;   // ... of the B operand (TN / NWV rows per wave)
  constexpr int RPP = 64 / KCW;  // rows per piece
  using C           = Chunk<KCW>;
  constexpr int STAGE = (TMV + TN) * C::ROWBYTES;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* sA  = smem;
  char* sB  = smem + TMV * C::ROWBYTES;
  int*  pcA = reinterpret_cast<int*>(smem + (PIPE ? PIPE : 1) * STAGE);
  int*  pcB = pcA + TMV;

  // Workgroup -> tile map: blockIdx.y walks 64 x 64-tile supertiles, blockIdx.x walks a supertile with
  // tile_n fastest.  Inside a supertile both operand blocks (8 MB each) stay in L2 / Infinity Cache, so only
  // the first touch of a row block pays HBM latency (a 1M-row operand streamed tile by tile made EVERY chunk
  // load a first touch: 0.41 vs 0.58 T pairs/s).  Block b runs on XCD b % 8, so each XCD's L2 keeps 8 of the
  // 64 B tiles of the supertile (1 MB) while the A tile is shared by 64 consecutive workgroups.
  const unsigned superN = (tilesN + SUPER - 1) / SUPER;
  const unsigned sm     = blockIdx.y / superN;
  const unsigned sn     = blockIdx.y - sm * superN;
  const unsigned tile_m = sm * SUPER + blockIdx.x / SUPER;
  const unsigned tile_n = sn * SUPER + (blockIdx.x & (SUPER - 1));
  if (tile_m >= tilesM || tile_n >= tilesN) return;

  const int     tid   = threadIdx.x;
  const int     lane  = tid & 63;
  const int     wave  = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int     wm    = wave >> 1;
  const int     wn    = wave & 1;
  const int64_t rowA0 = static_cast<int64_t>(tile_m) * TMV;
  const int64_t rowB0 = static_cast<int64_t>(tile_n) * TN;

  static_assert(PIPE == 0 || TMV == TM, "the ring variant is written for the 128 x 128 tile");
  if (tid < TMV) {
    pcA[tid] = popA[rowA0 + tid];
  } else if (tid - TMV < TN) {
    pcB[tid - TMV] = popB[rowB0 + tid - TMV];
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
    const uint4*   gB      = B + (rowB0 + prowB) * Wp;
    const int      nChunks = Wp / KCW;
    if constexpr (PIPE != 0) {
      // DMA through inline asm (invisible to the compiler's waitcnt pass, see dma_b128) + counted waits: chunk ch + 1
      // and ch + 2 are in flight while chunk ch is multiplied; one barrier per chunk.
      unsigned offA[PPW], offB[PPW];
#pragma unroll
      for (int t = 0; t < PPW; ++t) {
        const unsigned slot = (static_cast<unsigned>(lane) & (KCW - 1u)) ^ C::swz(prow + RPP * t);
        offA[t] = offB[t]   = static_cast<unsigned>(t * rowStep) + slot;
      }
      auto issue = [&](const int ch) {
        char* st = smem + (ch % PIPE) * STAGE;
#pragma unroll
        for (int t = 0; t < PPW; ++t) dma_b128(gA + ch * KCW + offA[t], lds_addr(st + (wave * PPW + t) * 1024));
#pragma unroll
        for (int t = 0; t < PPW; ++t) dma_b128(gB + ch * KCW + offB[t], lds_addr(st + TMV * C::ROWBYTES + (wave * PPW + t) * 1024));
      };
      lds_barrier();  // popcounts are in LDS (their loads are the compiler's own and already waited for)
      for (int ch = 0; ch < PIPE - 1 && ch < nChunks; ++ch) issue(ch);
      for (int ch = 0; ch < nChunks; ++ch) {
        const int ahead = nChunks - 1 - ch < PIPE - 2 ? nChunks - 1 - ch : PIPE - 2;  // chunks issued after ch and still flying
        if (ahead >= 1) wait_vmcnt<2 * PPW>(); else wait_vmcnt<0>();
        lds_barrier();
        if (ch + PIPE - 1 < nChunks) issue(ch + PIPE - 1);
        char* st = smem + (ch % PIPE) * STAGE;
        chunk_mma<KCW>(acc, st, st + TMV * C::ROWBYTES, wm, wn, lane);
      }
    } else
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
        __builtin_amdgcn_global_load_lds((gptr_t)(gB + t * rowStep + ch * KCW + slot),
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
  const bool     full    = (rowA0 + TMV <= nA) && (rowB0 + TN <= nB);
  const unsigned hi      = static_cast<unsigned>(lane >> 5);
  const unsigned laneOff = (hi * 4u * static_cast<unsigned>(ld) + static_cast<unsigned>(lane & 31)) * 8u;
  char*          waveOut = reinterpret_cast<char*>(out + (rowA0 + wm * 64) * ld + rowB0 + wn * 64);
  // popcounts and the f32 accumulators are exact integers < 2^24, so the union is formed in f32 (no int round trip)
  auto value = [&](const float c, const float pav, const float pbv) -> double {
    if constexpr (METRIC == NVMK_METRIC_TANIMOTO) {
      const float u = pav + pbv - c;
      return ratio_by_newton_f(c, fmaxf(u, 1.0f));
    } else {
      const double denom = sqrt(static_cast<double>(pav) * static_cast<double>(pbv));
      return (c == 0.0f || denom == 0.0) ? 0.0 : static_cast<double>(c) / denom;
    }
  };
  const float pb0 = static_cast<float>(pcB[wn * 64 + (lane & 31)]);
  const float pb1 = static_cast<float>(pcB[wn * 64 + 32 + (lane & 31)]);
  auto emit = [&](auto fullTag) {
    constexpr bool FULL = decltype(fullTag)::value;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int   il0    = mi * 32 + (r & 3) + 8 * (r >> 2);  // + 4 hi: this lane's row inside the wave tile
        const float pav    = static_cast<float>(pcA[wm * 64 + il0 + 4 * static_cast<int>(hi)]);
        char*       rowOut = waveOut + static_cast<int64_t>(il0) * ld * 8;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const double v   = value(acc[mi][ni][r], pav, ni ? pb1 : pb0);
          double*      dst = reinterpret_cast<double*>(rowOut + ni * 256 + laneOff);
          if (FULL || (rowA0 + wm * 64 + il0 + 4 * static_cast<int>(hi) < nA &&
                       rowB0 + wn * 64 + ni * 32 + (lane & 31) < nB)) {
            // nontemporal: the 8 B/pair output stream must not evict the operand blocks from L2
            __builtin_nontemporal_store(v, dst);
          }
        }
      }
    }
  };
  // Interior tiles: neighbouring lanes swap one value (DPP quad_perm, no LDS) so that every lane owns two ADJACENT
  // columns of one row and the tile goes out as 32 sixteen-byte stores per lane instead of 64 eight-byte ones —
  // half the vector-memory instructions queued in front of the other workgroups' operand loads.  Accumulator rows
  // r = 4 q + s are tile rows s + 8 q: rows s and s + 1 pair up; even lanes keep row s, odd lanes row s + 1.
  auto emit_wide = [&]() {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const unsigned odd     = static_cast<unsigned>(lane) & 1u;
    const unsigned laneOf2 = (hi * 4u * static_cast<unsigned>(ld) + odd * static_cast<unsigned>(ld) + (static_cast<unsigned>(lane & 31) & ~1u)) * 8u;
    auto swap1 = [&](const double x) -> double {  // value of lane ^ 1
      const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xB1, 0xF, 0xF, false);
      const int hh = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xB1, 0xF, 0xF, false);
      return __hiloint2double(hh, lo);
    };
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int sp = 0; sp < 4; sp += 2) {
          const int   il0  = mi * 32 + sp + 8 * q;  // row of the even lanes (+ 4 hi); odd lanes: il0 + 1
          const float pav0 = static_cast<float>(pcA[wm * 64 + il0 + 4 * static_cast<int>(hi)]);
          const float pav1 = static_cast<float>(pcA[wm * 64 + il0 + 1 + 4 * static_cast<int>(hi)]);
          char*       rowOut = waveOut + static_cast<int64_t>(il0) * ld * 8;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const double v0   = value(acc[mi][ni][4 * q + sp], pav0, ni ? pb1 : pb0);      // (row s,     my column)
            const double v1   = value(acc[mi][ni][4 * q + sp + 1], pav1, ni ? pb1 : pb0);  // (row s + 1, my column)
            const double got  = swap1(odd ? v0 : v1);                                      // partner's value for MY row
            d2_t         out2;
            out2.x = odd ? got : v0;
            out2.y = odd ? v1 : got;
            __builtin_nontemporal_store(out2, reinterpret_cast<d2_t*>(rowOut + ni * 256 + laneOf2));
          }
        }
      }
    }
  };
  if (full) {
    emit_wide();
  } else {
    emit(std::false_type{});
  }
}

// ---- helpers of the producer / consumer kernel below ----------------------------------------------
constexpr int BN = 192;  // tile columns

// ---- dense cross-similarity, producer / consumer kernel --------------------------------------------
// One persistent 1024-thread workgroup per CU, no workgroup barrier after start-up:
//   * 4 loader waves stream the operand stages (KW words of 128 + 192 rows, LDS-DMA) of a SEQUENCE of 128 x 192 tiles
//     into a ring of LDS slots, always D = STAGES - 2 stages in flight, and publish "slot s holds its k-th stage" by
//     bumping readyCnt[s] after a counted s_waitcnt vmcnt;
//   * two groups of 6 compute waves (2 x 3, 64 x 64 each) take the tiles alternately: poll readyCnt, ds_read + MFMA,
//     bump freeCnt[s] (the loaders poll it before they overwrite a slot), and after the last stage of their tile run
//     the whole 64-element epilogue as one straight-line block.  While one group converts and stores tile T the other
//     multiplies tile T + 1, whose stages the loaders had already started fetching, so loads, matrix work and stores
//     of ONE workgroup overlap (in the 128 x 128 kernel they only overlap across co-resident workgroups, by luck).
// LDS operations of a wave execute in order, so a wave's freeCnt bump cannot pass its ds_reads of that slot, and a
// loader's readyCnt bump cannot pass the DMA writes it waited for.  Loaders never store and compute waves never load
// from global memory, so nobody's vmcnt mixes loads and stores.
constexpr int PM = 128;  // tile rows

__device__ __forceinline__ int lds_peek(const int* p) {  // ds_read the compiler may neither cache nor reorder
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p))) : "memory");
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void lds_bump(int* p, const int lane) {
  if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p))), "v"(1) : "memory");
}

template <int METRIC, int STAGES, int KW, bool PROF = false>
__global__ __launch_bounds__(1024, 1) void cross_sim_pp_kernel(const uint4* __restrict__ A, const int32_t* __restrict__ popA,
                                                            const int64_t nA, const uint4* __restrict__ B,
                                                            const int32_t* __restrict__ popB, const int64_t nB, const int Wp,
                                                            double* __restrict__ out, const int64_t ld, const unsigned tilesM,
                                                            const unsigned tilesN, long long* __restrict__ prof) {
  using C = Chunk<KW>;
  long long nPollFail = 0, tEpi = 0, tAll = PROF ? static_cast<long long>(wall_clock64()) : 0;
  constexpr int STAGE_A = PM * KW * 16;                // 8 KB at KW = 4
  constexpr int STAGE   = STAGE_A + BN * KW * 16;      // 20 KB at KW = 4
  constexpr int RP      = 64 / KW;                     // rows one DMA instruction covers
  constexpr int APL = PM / RP / 4, BPL = BN / RP / 4;  // DMA instructions per loader wave and stage, A and B
  constexpr int LOADS   = APL + BPL;
  constexpr int D       = STAGES - 2;                  // stages in flight
  constexpr int PCBUF   = PM + BN;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  int* pcBase   = reinterpret_cast<int*>(smem + STAGES * STAGE);  // [4][PM + BN], by tile index & 3
  int* readyCnt = pcBase + 4 * PCBUF;                             // [STAGES]
  int* freeCnt  = readyCnt + STAGES;                              // [STAGES]

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nst  = Wp / KW;
  // Tile order: 16 x 16-tile supertiles, row-major inside (tile column fastest).  The workgroups of one step work
  // through the same supertile, so the chip writes a 24 KB wide stripe of the output at any moment and XCD x
  // (workgroups = x mod 8) keeps tile columns x and x + 8 of the supertile in its L2.  Slots past the matrix edge
  // are computed on clamped operands and stored nowhere.
  constexpr unsigned SUP = 16;
  const unsigned superM = (tilesM + SUP - 1) / SUP, superN = (tilesN + SUP - 1) / SUP;
  const unsigned long long nSlots = static_cast<unsigned long long>(superM) * superN * (SUP * SUP);
  const unsigned long long first = blockIdx.x, stride = gridDim.x;
  if (first >= nSlots) return;
  const int myTiles     = static_cast<int>((nSlots - first + stride - 1) / stride);
  const int totalStages = myTiles * nst;
  auto tile_of = [&](const int k, unsigned& tm, unsigned& tn) {
    const unsigned long long t = first + static_cast<unsigned long long>(k) * stride;
    const unsigned sidx = static_cast<unsigned>(t / (SUP * SUP)), within = static_cast<unsigned>(t % (SUP * SUP));
    const unsigned sm = sidx / superN, sn = sidx - sm * superN;
    tm = sm * SUP + within / SUP;
    tn = sn * SUP + within % SUP;
  };
  if (tid < 2 * STAGES) readyCnt[tid] = 0;
  __syncthreads();

  if (wave >= 12) {
    // ---------------- loader waves ----------------
    const int q = wave - 12;
    // Loaders outrank the compute waves they share a SIMD with: their few address instructions must not queue behind
    // the epilogue's f64 stream, or the DMA pipe runs dry.
    __builtin_amdgcn_s_setprio(3);
    unsigned offA[APL], offB[BPL];  // this lane's 16-byte piece inside the tile, in uint4 units (without the stage's word offset)
#pragma unroll
    for (int k = 0; k < APL; ++k) {
      const unsigned row = static_cast<unsigned>((q * APL + k) * RP + lane / KW);
      offA[k]            = row * static_cast<unsigned>(Wp) + ((static_cast<unsigned>(lane) & (KW - 1)) ^ C::swz(row));
    }
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
      const unsigned row = static_cast<unsigned>((q * BPL + k) * RP + lane / KW);
      offB[k]            = row * static_cast<unsigned>(Wp) + ((static_cast<unsigned>(lane) & (KW - 1)) ^ C::swz(row));
    }
    // issue cursor: stage gi = tile ti, word chunk ci, ring slot si, its use count ui
    int          gi = 0, ti = 0, ci = 0, si = 0, ui = 0;
    const uint4 *gA = nullptr, *gB = nullptr;
    const int32_t *pA = nullptr, *pB = nullptr;
    auto issue = [&]() {
      if (ci == 0) {
        unsigned tm, tn;
        tile_of(ti, tm, tn);
        tm = tm < tilesM ? tm : tilesM - 1;
        tn = tn < tilesN ? tn : tilesN - 1;
        gA = A + static_cast<int64_t>(tm) * PM * Wp;
        gB = B + static_cast<int64_t>(tn) * BN * Wp;
        pA = popA + static_cast<int64_t>(tm) * PM;
        pB = popB + static_cast<int64_t>(tn) * BN;
      }
      if (ui > 0) {  // the slot's previous stage must have been read by its six consumers
        while (lds_peek(freeCnt + si) < 6 * ui) { __builtin_amdgcn_s_sleep(1); if constexpr (PROF) ++nPollFail; }
      }
      char*        st = smem + si * STAGE;
      const uint4* sa = gA + ci * KW;  // wave-uniform bases; the per-lane part is a 32-bit offset
      const uint4* sb = gB + ci * KW;
#pragma unroll
      for (int k = 0; k < APL; ++k) {
        __builtin_amdgcn_global_load_lds((gptr_t)(sa + offA[k]), (lptr_t)(st + (q * APL + k) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < BPL; ++k) {
        __builtin_amdgcn_global_load_lds((gptr_t)(sb + offB[k]), (lptr_t)(st + STAGE_A + (q * BPL + k) * 1024), 16, 0, 0);
      }
      if (ci == 0) {  // popcounts: two dword DMA loads per loader wave (repeats keep the count equal on every wave)
        int*      pc = pcBase + (ti & 3) * PCBUF;
        const int pa = q & 1, pb = q < 3 ? q : 0;
        __builtin_amdgcn_global_load_lds((gptr_t)(pA + pa * 64 + lane), (lptr_t)(pc + pa * 64), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(pB + pb * 64 + lane), (lptr_t)(pc + PM + pb * 64), 4, 0, 0);
      }
      ++gi;
      if (++ci == nst) { ci = 0; ++ti; }
      if (++si == STAGES) { si = 0; ++ui; }
    };
    int ce = 0, se = 0;  // oldest unpublished stage: chunk index and slot
    for (int g = 0; g < totalStages + D; ++g) {
      if (g >= D) {  // publish stage e = g - D: everything this wave issued for it has landed
        const int e = g - D;
        int younger = 0, cc = ce;
#pragma unroll
        for (int k = 1; k < D; ++k) {
          if (++cc == nst) cc = 0;
          if (e + k < gi) younger += (cc == 0) ? LOADS + 2 : LOADS;
        }
        wait_vmcnt_dyn(younger);
        lds_bump(readyCnt + se, lane);
        if (++ce == nst) ce = 0;
        if (++se == STAGES) se = 0;
      }
      if (g < totalStages) issue();
    }
    if constexpr (PROF) {
      if (blockIdx.x == 0 && lane == 0) { prof[wave * 4] = nPollFail; prof[wave * 4 + 1] = 0; prof[wave * 4 + 2] = static_cast<long long>(wall_clock64()) - tAll; prof[wave * 4 + 3] = totalStages; }
    }
    return;
  }

  // ---------------- compute waves ----------------
  const int      group = wave / 6, w6 = wave % 6;
  const int      wm = w6 / 3, wn = w6 % 3;
  const int      l31  = lane & 31;
  const unsigned half = static_cast<unsigned>(lane >> 5);
  const unsigned rA   = static_cast<unsigned>(wm * 64 + l31), rB = static_cast<unsigned>(wn * 64 + l31);
  const unsigned baseA = rA * C::ROWBYTES, baseB = STAGE_A + rB * C::ROWBYTES;
  const unsigned swA = C::swz(rA), swB = C::swz(rB);
  for (int T = group; T < myTiles; T += 2) {
    v16f acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
      }
    }
    int g    = T * nst;
    int slot = g % STAGES, need = 4 * (g / STAGES + 1);
    __builtin_amdgcn_s_setprio(2);  // the group feeding the matrix cores goes before the group converting and storing
    for (int c = 0; c < nst; ++c) {
      while (lds_peek(readyCnt + slot) < need) { __builtin_amdgcn_s_sleep(1); if constexpr (PROF) ++nPollFail; }
      const char* st = smem + slot * STAGE;
#pragma unroll
      for (int ks = 0; ks < KW / 2; ++ks) {
        const unsigned sl   = static_cast<unsigned>(ks * 2) + half;
        const unsigned offA = baseA + ((sl ^ swA) << 4), offB = baseB + ((sl ^ swB) << 4);
        const uint4    a0 = *reinterpret_cast<const uint4*>(st + offA);
        const uint4    a1 = *reinterpret_cast<const uint4*>(st + offA + 32 * C::ROWBYTES);
        const uint4    b0 = *reinterpret_cast<const uint4*>(st + offB);
        const uint4    b1 = *reinterpret_cast<const uint4*>(st + offB + 32 * C::ROWBYTES);
        acc[0][0]         = mfma_fp4(a0, b0, acc[0][0]);
        acc[0][1]         = mfma_fp4(a0, b1, acc[0][1]);
        acc[1][0]         = mfma_fp4(a1, b0, acc[1][0]);
        acc[1][1]         = mfma_fp4(a1, b1, acc[1][1]);
      }
      asm volatile("" ::: "memory");
      lds_bump(freeCnt + slot, lane);  // queued behind this wave's ds_reads of the slot
      if (++slot == STAGES) { slot = 0; need += 4; }
    }
    // epilogue (same element order as the 128 x 128 kernel), one straight-line block
    __builtin_amdgcn_s_setprio(0);
    const long long te0 = PROF ? static_cast<long long>(wall_clock64()) : 0;
    unsigned tm, tn;
    tile_of(T, tm, tn);
    const int*     pc      = pcBase + (T & 3) * PCBUF;
    const int64_t  rowA0   = static_cast<int64_t>(tm) * PM, rowB0 = static_cast<int64_t>(tn) * BN;
    const bool     full    = (rowA0 + PM <= nA) && (rowB0 + BN <= nB);
    const unsigned laneOff = (half * 4u * static_cast<unsigned>(ld) + static_cast<unsigned>(l31)) * 8u;
    char*          waveOut = reinterpret_cast<char*>(out + (rowA0 + wm * 64) * ld + rowB0 + wn * 64);
    const int      pb0 = pc[PM + wn * 64 + l31], pb1 = pc[PM + wn * 64 + 32 + l31];
    auto value = [&](const int cnt, const int pav, const int pbv) -> double {
      if constexpr (METRIC == NVMK_METRIC_TANIMOTO) {
        const int u = pav + pbv - cnt;
        return ratio_by_newton(cnt, u > 1 ? u : 1);
      } else {
        const double denom = sqrt(static_cast<double>(pav) * static_cast<double>(pbv));
        return (cnt == 0 || denom == 0.0) ? 0.0 : static_cast<double>(cnt) / denom;
      }
    };
    auto emit = [&](auto fullTag) {
      constexpr bool FULL = decltype(fullTag)::value;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int il0    = mi * 32 + (r & 3) + 8 * (r >> 2);
          const int pav    = pc[wm * 64 + il0 + 4 * static_cast<int>(half)];
          char*     rowOut = waveOut + static_cast<int64_t>(il0) * ld * 8;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const double v   = value(static_cast<int>(acc[mi][ni][r]), pav, ni ? pb1 : pb0);
            double*      dst = reinterpret_cast<double*>(rowOut + ni * 256 + laneOff);
            if (FULL || (rowA0 + wm * 64 + il0 + 4 * static_cast<int>(half) < nA && rowB0 + wn * 64 + ni * 32 + l31 < nB)) {
              __builtin_nontemporal_store(v, dst);
            }
          }
        }
      }
    };
    if (full) {
      emit(std::true_type{});
    } else {
      emit(std::false_type{});
    }
    if constexpr (PROF) tEpi += static_cast<long long>(wall_clock64()) - te0;
  }
  if constexpr (PROF) {
    if (blockIdx.x == 0 && lane == 0) { prof[wave * 4] = nPollFail; prof[wave * 4 + 1] = tEpi; prof[wave * 4 + 2] = static_cast<long long>(wall_clock64()) - tAll; prof[wave * 4 + 3] = totalStages; }
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
  const unsigned long long edgeCapacity, const double K1, const double K2, const double adj, const float bandThr) {
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
    const double S = static_cast<double>(superN);
    unsigned     r = static_cast<unsigned>((2.0 * S + 1.0 - sqrt((2.0 * S + 1.0) * (2.0 * S + 1.0) - 8.0 * static_cast<double>(sidx))) * 0.5);
    auto rowStart  = [&](const unsigned q) { return static_cast<unsigned long long>(q) * (2ull * superN - q + 1ull) / 2ull; };
    while (r > 0 && rowStart(r) > sidx) --r;
    while (rowStart(r + 1) <= sidx) ++r;
    sm = r;
    sn = r + static_cast<unsigned>(sidx - rowStart(r));
    if (sm >= superN) return;  // padding of the 2-D supertile grid
  } else {
    sm = sidx / superN;
    sn = sidx - sm * superN;
  }
  const unsigned tile_m = sm * superH + blockIdx.x / superW;
  const unsigned tile_n = sn * superW + (blockIdx.x - (blockIdx.x / superW) * superW);
  if (tile_m >= tilesM || tile_n >= tilesN) return;
  if (symmetric && tile_n < tile_m) return;
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

// ---- neighbour counting, ring kernel (un-gathered all-pairs passes) ------------------------------
// The tile kernel above spends its time in phases that do not overlap inside a workgroup and pulls 16 B of operand
// bytes per pair through L2.  For the big passes (fused Butina's all-pairs pass, nvmk_neighbor_counts) this kernel
// is a conventional pipelined GEMM instead — possible here because there is no output stream to drain:
//   * ONE persistent 512-thread workgroup per CU, tile 256 x 256 (8 B/pair), 8 waves of 64 x 128 (6 ds_read_b128 per
//     8 MFMAs instead of 4 per 4);
//   * a ring of four 4-word stages (32 KB each): every wave issues its share of LDS-DMA loads three stages ahead and
//     waits with a COUNTED s_waitcnt vmcnt (wave-uniform bookkeeping of everything it has issued, so the two count
//     atomics per tile are accounted for), one s_barrier per stage; the stage stream runs on across tile boundaries,
//     so the next tile's operands arrive under the epilogue;
//   * tiles are walked in 16 x 16-tile superblocks, each XCD (workgroups = x mod 8) owning a 4 x 8 sub-block so its
//     L2 holds 4 + 8 operand tiles; symmetric passes enumerate only the superblocks on or above the diagonal;
//   * neighbour pairs are staged in LDS and flushed in bulk (one returning atomic per ~512 pairs, not per ballot).
constexpr int RT  = 256;  // tile edge
constexpr int RKW = 4;    // words per stage
constexpr int RST = 4;    // ring slots
constexpr int REDGE_CAP = 1024, REDGE_FLUSH = 512;

template <int METRIC, bool EMIT, bool PROF = false, int WT = 128>
__global__ __launch_bounds__(64 * 4 * (RT / WT), 1) void neighbor_count_ring_kernel(
  const uint4* __restrict__ X, const int32_t* __restrict__ popX, const int64_t nX, const uint4* __restrict__ Y,
  const int32_t* __restrict__ popY, const int64_t nY, const int Wp, const double K1, const double K2, const double adj, const float thr,
  const int sign, const int symmetric, int32_t* __restrict__ counts, int2* __restrict__ edges,
  unsigned long long* __restrict__ edgeCursor, const unsigned long long edgeCapacity, long long* __restrict__ prof) {
  using C = Chunk<RKW>;
  auto      now = [&]() -> long long { return PROF ? static_cast<long long>(wall_clock64()) : 0; };
  long long tWait = 0, tMma = 0, tEpi = 0, nTiles = 0;
  constexpr int STAGE_X = RT * RKW * 16;  // 16 KB
  constexpr int STAGE   = 2 * STAGE_X;    // 32 KB
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  int*  pcBase   = reinterpret_cast<int*>(smem + RST * STAGE);  // [2][512]: tile parity, X rows then Y rows
  int*  rowsum   = pcBase + 2 * 2 * RT;                          // [256]
  int*  colsum   = rowsum + RT;                                  // [256]
  int2* edgeBuf  = reinterpret_cast<int2*>(colsum + RT);         // [REDGE_CAP]
  int*  edgeMeta = reinterpret_cast<int*>(edgeBuf + REDGE_CAP);  // [0] staged count, [2..3] flush base

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 4 x RWN waves of 64 x WT: WT = 128 -> 8 waves (6 ds_read_b128 per 8 MFMAs, 2 waves per SIMD), WT = 64 -> 16 waves
  // (4 per 4, 4 waves per SIMD)
  constexpr int RWN = RT / WT, NWV = 4 * RWN, NI = WT / 32, RNTV = 64 * NWV, PPV = 16 / NWV;
  const int wm = wave / RWN, wn = wave % RWN;
  const int nst  = Wp / RKW;
  constexpr int INVALID = (METRIC == NVMK_METRIC_TANIMOTO) ? -1 : 0;  // popcount stand-in of rows that can have no neighbour
  const unsigned tilesM = static_cast<unsigned>((nX + RT - 1) / RT), tilesN = static_cast<unsigned>((nY + RT - 1) / RT);
  constexpr unsigned SUP = 16;
  const unsigned superM = (tilesM + SUP - 1) / SUP, superN = (tilesN + SUP - 1) / SUP;
  const unsigned long long nSuper = symmetric ? static_cast<unsigned long long>(superN) * (superN + 1ull) / 2ull :
                                                static_cast<unsigned long long>(superM) * superN;
  // slot k of this workgroup -> tile; false when the slot holds no work (edge of the matrix, below the diagonal)
  auto tile_at = [&](const long long k, unsigned& tm, unsigned& tn, bool& end) -> bool {
    const unsigned long long t  = blockIdx.x + static_cast<unsigned long long>(k) * gridDim.x;
    const unsigned long long sb = t / (SUP * SUP);
    end                         = sb >= nSuper;
    if (end) return false;
    const unsigned within = static_cast<unsigned>(t % (SUP * SUP));
    unsigned       sm, sn;
    if (symmetric) {
      const double S = static_cast<double>(superN);
      unsigned     r = static_cast<unsigned>((2.0 * S + 1.0 - sqrt((2.0 * S + 1.0) * (2.0 * S + 1.0) - 8.0 * static_cast<double>(sb))) * 0.5);
      auto rowStart  = [&](const unsigned q) { return static_cast<unsigned long long>(q) * (2ull * superN - q + 1ull) / 2ull; };
      while (r > 0 && rowStart(r) > sb) --r;
      while (rowStart(r + 1) <= sb) ++r;
      sm = r;
      sn = r + static_cast<unsigned>(sb - rowStart(r));
    } else {
      sm = static_cast<unsigned>(sb / superN);
      sn = static_cast<unsigned>(sb - static_cast<unsigned long long>(sm) * superN);
    }
    const unsigned x = within & 7u, j = within >> 3;  // XCD x owns a 4 x 8 sub-block of the superblock
    tm = sm * SUP + (x >> 1) * 4u + (j >> 3);
    tn = sn * SUP + (x & 1u) * 8u + (j & 7u);
    return tm < tilesM && tn < tilesN && (!symmetric || tn >= tm);
  };
  auto next_tile = [&](long long& k, unsigned& tm, unsigned& tn) -> bool {  // advance to the next slot with work
    bool end = false;
    for (++k;; ++k) {
      if (tile_at(k, tm, tn, end)) return true;
      if (end) return false;
    }
  };

  if (tid < RT) { rowsum[tid] = 0; colsum[tid] = 0; }
  if (tid == 0) edgeMeta[0] = 0;
  lds_barrier();

  // ---- load side: the stage stream of this workgroup's tile sequence ----
  long long kIss = -1;
  unsigned  itm = 0, itn = 0;
  bool      issLive = next_tile(kIss, itm, itn);
  int       ici = 0, islot = 0, itile = 0;  // chunk inside the tile, ring slot, ordinal of the tile in the sequence
  int       issuedTotal = 0;                // vector-memory operations this wave has issued so far
  int       issuedAt[RST] = {0, 0, 0, 0};   // ... right after the loads of the stage in each slot
  const unsigned prow0 = static_cast<unsigned>(wave * PPV * 16 + (lane >> 2));  // rows of this wave's PPV 16-row pieces
  unsigned       offP[PPV];
#pragma unroll
  for (int t = 0; t < PPV; ++t) {
    const unsigned row = prow0 + 16u * t;
    offP[t]            = row * static_cast<unsigned>(Wp) + ((static_cast<unsigned>(lane) & 3u) ^ C::swz(row));
  }
  auto issue_stage = [&]() {
    if (issLive) {
      char*        st = smem + islot * STAGE;
      const uint4* gx = X + static_cast<int64_t>(itm) * RT * Wp + ici * RKW;
      const uint4* gy = Y + static_cast<int64_t>(itn) * RT * Wp + ici * RKW;
#pragma unroll
      for (int t = 0; t < PPV; ++t) {
        dma_b128(gx + offP[t], lds_addr(st + (wave * PPV + t) * 1024));
      }
#pragma unroll
      for (int t = 0; t < PPV; ++t) {
        dma_b128(gy + offP[t], lds_addr(st + STAGE_X + (wave * PPV + t) * 1024));
      }
      issuedTotal += 2 * PPV;
      if (ici == 0 && wave < 8) {  // row popcounts of the tile: 512 dwords, one DMA instruction on waves 0-3 (X rows) and 4-7 (Y rows)
        int*           pc  = pcBase + (itile & 1) * 2 * RT;
        const int32_t* src = wave < 4 ? popX + static_cast<int64_t>(itm) * RT + wave * 64 : popY + static_cast<int64_t>(itn) * RT + (wave - 4) * 64;
        dma_b32(src + lane, lds_addr(pc + wave * 64));
        issuedTotal += 1;
      }
      issuedAt[islot] = issuedTotal;
      if (++ici == nst) {
        ici = 0;
        ++itile;
        issLive = next_tile(kIss, itm, itn);
      }
    } else {
      issuedAt[islot] = issuedTotal;
    }
    if (++islot == RST) islot = 0;
  };
  for (int i = 0; i < RST - 1; ++i) issue_stage();

  // ---- compute side ----
  const int      l31  = lane & 31;
  const unsigned half = static_cast<unsigned>(lane >> 5);
  const unsigned rA   = static_cast<unsigned>(wm * 64 + l31), rB = static_cast<unsigned>(wn * WT + l31);
  const unsigned baseA = rA * C::ROWBYTES, baseB = STAGE_X + rB * C::ROWBYTES;
  const unsigned swA = C::swz(rA), swB = C::swz(rB);
  long long kCmp = -1;
  unsigned  tm = 0, tn = 0;
  int       cslot = 0, ctile = 0;
  while (next_tile(kCmp, tm, tn)) {
    v16f acc[2][NI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
      }
    }
    const int64_t rowA0 = static_cast<int64_t>(tm) * RT, rowB0 = static_cast<int64_t>(tn) * RT;
    int*          pc    = pcBase + (ctile & 1) * 2 * RT;
    for (int ch = 0; ch < nst; ++ch) {
      const long long t0 = now();
      wait_vmcnt_dyn(issuedTotal - issuedAt[cslot]);  // this wave's loads of the stage have landed
      if (ch == 0 && METRIC == NVMK_METRIC_TANIMOTO) {
        // rows past the end and empty fingerprints (union 0 -> similarity 0 -> never >= a positive threshold) never
        // match: flag them (this thread DMA'd the entry it patches, and its wait has just completed)
        if (tid < 2 * RT) {
          const int64_t r = tid < RT ? rowA0 + tid : rowB0 + (tid - RT);
          if (r >= (tid < RT ? nX : nY) || pc[tid] == 0) pc[tid] = INVALID;
        }
      }
      __builtin_amdgcn_s_barrier();  // ... and everyone else's; everyone is also done reading the slot refilled next
      asm volatile("" ::: "memory");
      const long long t1 = now();
      issue_stage();
      const char* st = smem + cslot * STAGE;
#pragma unroll
      for (int ks = 0; ks < RKW / 2; ++ks) {
        const unsigned sl   = static_cast<unsigned>(ks * 2) + half;
        const unsigned offA = baseA + ((sl ^ swA) << 4), offB = baseB + ((sl ^ swB) << 4);
        uint4 a[2], b[NI];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const uint4*>(st + offA + i * 32 * C::ROWBYTES);
#pragma unroll
        for (int jn = 0; jn < NI; ++jn) b[jn] = *reinterpret_cast<const uint4*>(st + offB + jn * 32 * C::ROWBYTES);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int jn = 0; jn < NI; ++jn) acc[i][jn] = mfma_fp4(a[i], b[jn], acc[i][jn]);
        }
      }
      asm volatile("" ::: "memory");
      if constexpr (PROF) {
        asm volatile("s_nop 0" : "+v"(acc[0][0]), "+v"(acc[1][NI - 1]));
        tWait += t1 - t0;
        tMma += now() - t1;
      }
      if (++cslot == RST) cslot = 0;
    }
    const long long te = now();

    // ---- epilogue: threshold, row / column counts, staged neighbour pairs ----
    const bool creditCols = symmetric && tn > tm;
    int        cc[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) cc[ni] = 0;
    int        myRow = 0;
    // Tanimoto: the table-free exact predicate with the row-slot fast reject (see ArithThreshold above)
    int    pbv[NI];
    double pbK[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      pbv[ni] = pc[RT + wn * WT + ni * 32 + l31];
      pbK[ni] = pbv[ni] < 0 ? __builtin_inf() : static_cast<double>(pbv[ni]) * K2;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowLo = mi * 32 + (r & 3) + 8 * (r >> 2);
        const int pav   = pc[wm * 64 + rowLo + 4 * static_cast<int>(half)];
        double    d[NI], paK = 0.0;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) d[ni] = 0.0;
        if constexpr (METRIC == NVMK_METRIC_TANIMOTO) {
          paK = pav < 0 ? __builtin_inf() : __builtin_fma(static_cast<double>(pav), K2, -adj);
          double best = -__builtin_inf();
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            d[ni] = __builtin_fma(static_cast<double>(acc[mi][ni][r]), K1, -pbK[ni]);
            best  = fmax(best, d[ni]);
          }
          if (__ballot(best > paK) == 0) continue;  // no neighbour in this row slot
        }
        int lo = 0, hi = 0;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          bool p;
          if constexpr (METRIC == NVMK_METRIC_TANIMOTO) {
            p = d[ni] > paK;
          } else {
            p = cosine_neighbor(static_cast<int>(acc[mi][ni][r]), pav, pbv[ni], thr);
          }
          cc[ni] += p ? 1 : 0;
          const uint64_t m = __ballot(p);
          lo += __popc(static_cast<unsigned>(m));
          hi += __popc(static_cast<unsigned>(m >> 32));
          if constexpr (EMIT) {
            if (m != 0) {  // rare: pairs i < j once; the diagonal tile holds both orientations and the self pairs
              const int64_t  gi = rowA0 + wm * 64 + rowLo + 4 * static_cast<int>(half);
              const int64_t  gj = rowB0 + wn * WT + ni * 32 + l31;
              const bool     e  = p && gi < gj && gi < nX && gj < nY;
              const uint64_t me = __ballot(e);
              if (me != 0) {
                const int first = __ffsll(static_cast<long long>(me)) - 1;
                int       base  = 0;
                if (lane == first) base = atomicAdd(&edgeMeta[0], __popcll(me));  // LDS
                base = __shfl(base, first);
                if (e) {
                  const int slot = base + __popcll(me & ((1ull << lane) - 1ull));
                  if (slot < REDGE_CAP) {
                    edgeBuf[slot] = make_int2(static_cast<int>(gi), static_cast<int>(gj));
                  } else {  // staging overflow (a tile full of neighbours): straight to memory
                    const unsigned long long gs = atomicAdd(edgeCursor, 1ull);
                    if (gs < edgeCapacity) edges[gs] = make_int2(static_cast<int>(gi), static_cast<int>(gj));
                  }
                }
              }
            }
          }
        }
        asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(myRow) : "s"(lo), "s"(rowLo));
        asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(myRow) : "s"(hi), "s"(rowLo + 4));
      }
    }
    if (myRow != 0) atomicAdd(&rowsum[wm * 64 + lane], myRow);
    if (creditCols) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        int v = cc[ni];
        v += __shfl_xor(v, 32);
        if (lane < 32 && v != 0) atomicAdd(&colsum[wn * WT + ni * 32 + lane], v);
      }
    }
    lds_barrier();
    if (tid < 2 * RT) {  // exactly one global atomic instruction per wave 0-7 and tile (adds of 0 included): exact vmcnt bookkeeping
      const bool    isRow = tid < RT;
      const int     t     = isRow ? tid : tid - RT;
      const int64_t r     = (isRow ? rowA0 : rowB0) + t;
      const int64_t n     = isRow ? nX : nY;
      int           v     = isRow ? rowsum[t] : colsum[t];
      if (isRow) rowsum[t] = 0; else colsum[t] = 0;
      if (r >= n || (!isRow && !creditCols)) v = 0;
      atomicAdd(&counts[r < n ? r : n - 1], sign * v);
    }
    if (wave < 8) issuedTotal += 1;
    if constexpr (EMIT) {
      const int staged = edgeMeta[0];  // same value in every thread: last written before the barrier above
      if (staged > REDGE_FLUSH) {
        const int nStaged = staged < REDGE_CAP ? staged : REDGE_CAP;
        if (tid == 0) {
          const unsigned long long base = atomicAdd(edgeCursor, static_cast<unsigned long long>(nStaged));
          edgeMeta[2] = static_cast<int>(base & 0xffffffffull);
          edgeMeta[3] = static_cast<int>(base >> 32);
        }
        lds_barrier();
        const unsigned long long base = (static_cast<unsigned long long>(static_cast<unsigned>(edgeMeta[3])) << 32) | static_cast<unsigned>(edgeMeta[2]);
        for (int i = tid; i < nStaged; i += RNTV) {
          if (base + i < edgeCapacity) edges[base + i] = edgeBuf[i];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // data-dependent number of stores: resynchronise the bookkeeping
#pragma unroll
        for (int q = 0; q < RST; ++q) issuedAt[q] = issuedTotal;
        lds_barrier();
        if (tid == 0) edgeMeta[0] = 0;
        lds_barrier();
      }
    }
    ++ctile;
    if constexpr (PROF) { tEpi += now() - te; ++nTiles; }
  }
  if constexpr (PROF) {
    if (blockIdx.x == 0 && lane == 0) { prof[wave * 4] = tWait; prof[wave * 4 + 1] = tMma; prof[wave * 4 + 2] = tEpi; prof[wave * 4 + 3] = nTiles; }
  }
  if constexpr (EMIT) {  // final flush
    lds_barrier();
    const int staged  = edgeMeta[0];
    const int nStaged = staged < REDGE_CAP ? staged : REDGE_CAP;
    if (nStaged > 0) {
      if (tid == 0) {
        const unsigned long long base = atomicAdd(edgeCursor, static_cast<unsigned long long>(nStaged));
        edgeMeta[2] = static_cast<int>(base & 0xffffffffull);
        edgeMeta[3] = static_cast<int>(base >> 32);
      }
      lds_barrier();
      const unsigned long long base = (static_cast<unsigned long long>(static_cast<unsigned>(edgeMeta[3])) << 32) | static_cast<unsigned>(edgeMeta[2]);
      for (int i = tid; i < nStaged; i += RNTV) {
        if (base + i < edgeCapacity) edges[base + i] = edgeBuf[i];
      }
    }
  }
}

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
                   unsigned tilesN, hipStream_t stream) {
  const size_t shmem = static_cast<size_t>(TM + TN) * 8 * 16 + (TM + TN) * 4;
  if (const char* pe = std::getenv("NVMK_DENSE_KERNEL"); pe != nullptr && std::string(pe) == "wide" && A.L.nPad % (2 * TM) == 0) {
    // experiment (0.58 vs 0.60 T pairs/s): 256 x 128 tiles, 8 waves, 2 workgroups per CU, 12 instead of 16 B/pair of
    // operand traffic.  Only when the row block is padded to whole 256-row tiles.
    constexpr int  TW  = 2 * TM;
    const unsigned tmw = static_cast<unsigned>(ceil_div<int64_t>(A.L.n, TW));
    const int64_t  sup = ceil_div<int64_t>(tmw, SUPER) * ceil_div<int64_t>(tilesN, SUPER);
    const size_t   shw = static_cast<size_t>(TW + TN) * 8 * 16 + (TW + TN) * 4;
    auto           kern = cross_sim_mfma_kernel<METRIC, 0, TW>;
    NVMK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shw)));
    hipLaunchKernelGGL(kern, dim3(grid.x, static_cast<unsigned>(sup)), dim3(2 * NT), shw, stream, A.rows, A.popc, A.L.n, B.rows, B.popc, B.L.n,
                       A.L.Wp, out, ld, tmw, tilesN);
    NVMK_LAUNCH_CHECK();
    return NVMK_OK;
  }
  if (const char* pe = std::getenv("NVMK_DENSE_KERNEL"); pe != nullptr && std::string(pe) == "pipe") {
    // experiment (0.56 vs 0.60 T pairs/s): three-stage ring of 4-word chunks inside the workgroup, 3 workgroups per CU
    const size_t shmem3 = static_cast<size_t>(3) * (TM + TN) * 4 * 16 + (TM + TN) * 4;
    hipLaunchKernelGGL((cross_sim_mfma_kernel<METRIC, 3>), grid, dim3(NT), shmem3, stream, A.rows, A.popc, A.L.n, B.rows, B.popc,
                       B.L.n, A.L.Wp, out, ld, tilesM, tilesN);
    NVMK_LAUNCH_CHECK();
    return NVMK_OK;
  }
  hipLaunchKernelGGL(cross_sim_mfma_kernel<METRIC>, grid, dim3(NT), shmem, stream, A.rows, A.popc, A.L.n, B.rows, B.popc,
                     B.L.n, A.L.Wp, out, ld, tilesM, tilesN);
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
  const int64_t supers = ceil_div<int64_t>(tilesM, SUPER) * ceil_div<int64_t>(tilesN, SUPER);
  NVMK_REQUIRE(supers <= 65535, "cross similarity: problem too large for one launch (%lld x %lld tiles)",
               (long long)tilesM, (long long)tilesN);
  const dim3     grid(static_cast<unsigned>(SUPER * SUPER), static_cast<unsigned>(supers));
  const unsigned tm = static_cast<unsigned>(tilesM), tn = static_cast<unsigned>(tilesN);
  // NVMK_DENSE_KERNEL=pp selects the experimental producer / consumer kernel (slower than the tile kernel as measured,
  // DESIGN.md §4.1; kept selectable because the parity tests run it and the next optimisation round starts from it).
  // It needs whole 4-word stages and operands zero-padded to 128- / 192-row tiles, which fp4::ROW_ALLOC guarantees.
  const char*       dk = std::getenv("NVMK_DENSE_KERNEL");
  const std::string dks(dk ? dk : "");
  if (dks == "pp" && A.L.Wp % 4 == 0) {
    static const int cus = [] {
      int dev = 0, n = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
      return n > 0 ? n : 256;
    }();
    {
      const unsigned btm = static_cast<unsigned>(ceil_div<int64_t>(A.L.n, PM)), btn = static_cast<unsigned>(ceil_div<int64_t>(B.L.n, BN));
      constexpr int PPS = 7, PKW = 4;  // 7 ring slots of 20 KB, 5 in flight
      const size_t  shmem = static_cast<size_t>(PPS) * (PM + BN) * PKW * 16 + 4 * (PM + BN) * 4 + 2 * PPS * 4;
      auto kern = (metric == NVMK_METRIC_TANIMOTO) ? cross_sim_pp_kernel<NVMK_METRIC_TANIMOTO, PPS, PKW> : cross_sim_pp_kernel<NVMK_METRIC_COSINE, PPS, PKW>;
      const bool profile = std::getenv("NVMK_PP_PROFILE") != nullptr && metric == NVMK_METRIC_TANIMOTO;
      if (profile) kern = cross_sim_pp_kernel<NVMK_METRIC_TANIMOTO, PPS, PKW, true>;
      long long* dProf = nullptr;
      if (profile) NVMK_HIP_CHECK(hipMalloc(&dProf, 64 * sizeof(long long)));
      NVMK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(shmem)));
      hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(cus)), dim3(1024), shmem, stream, A.rows, A.popc, A.L.n, B.rows, B.popc,
                         B.L.n, A.L.Wp, out, ld, btm, btn, dProf);
      NVMK_LAUNCH_CHECK();
      if (profile) {  // debugging aid: workgroup 0, per wave
        long long h[64];
        NVMK_HIP_CHECK(hipStreamSynchronize(stream));
        NVMK_HIP_CHECK(hipMemcpy(h, dProf, sizeof(h), hipMemcpyDeviceToHost));
        NVMK_HIP_CHECK(hipFree(dProf));
        for (int w = 0; w < 16; ++w)
          std::fprintf(stderr, "[pp] wave %2d %s: %lld stages, %.2f failed polls/stage, epilogue %.3f us/stage, total %.3f us/stage\n", w,
                       w < 12 ? "compute" : "loader ", h[w * 4 + 3], double(h[w * 4]) / h[w * 4 + 3], h[w * 4 + 1] * 0.01 / h[w * 4 + 3],
                       h[w * 4 + 2] * 0.01 / h[w * 4 + 3]);
      }
      return NVMK_OK;
    }
  }
  if (metric == NVMK_METRIC_TANIMOTO) return launch_dense_t<NVMK_METRIC_TANIMOTO>(A, B, out, ld, grid, tm, tn, stream);
  return launch_dense_t<NVMK_METRIC_COSINE>(A, B, out, ld, grid, tm, tn, stream);
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
  {
    // NVMK_COUNT_KERNEL=ring sends un-gathered passes to the pipelined ring kernel (256 x 256 tiles, one persistent
    // workgroup per CU).  Opt-in: measured 0.93 T pairs/s against 1.15 for the tile kernel.
    static const int cus = [] {
      int dev = 0, n = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
      return n > 0 ? n : 256;
    }();
    const char*       ck = std::getenv("NVMK_COUNT_KERNEL");
    const std::string cks(ck ? ck : "");
    const bool        plain = a.xRows == nullptr && a.yRows == nullptr && a.xIds == nullptr && a.yIds == nullptr && a.nXdev == nullptr &&
                       a.nYdev == nullptr && X.L.Wp % RKW == 0 && X.L.Wp / RKW >= RST && X.L.nPad % RT == 0 && Y.L.nPad % RT == 0 &&
                       std::max(X.L.nPad, Y.L.nPad) * X.L.Wp < (int64_t{1} << 32) &&
                       (a.metric != NVMK_METRIC_TANIMOTO || arith_threshold(a.thr, F).ok);
    if (plain && cks == "ring") {
      using RKern = void (*)(const uint4*, const int32_t*, int64_t, const uint4*, const int32_t*, int64_t, int, double, double, double, float, int,
                             int, int32_t*, int2*, unsigned long long*, unsigned long long, long long*);
      const ArithThreshold at = arith_threshold(a.thr, F);
      static const int wt = [] { const char* e = std::getenv("NVMK_RING_WT"); return (e && std::atoi(e) == 64) ? 64 : 128; }();
      RKern kern;
      if (a.metric == NVMK_METRIC_TANIMOTO) {
        kern = wt == 64 ? (emit ? neighbor_count_ring_kernel<NVMK_METRIC_TANIMOTO, true, false, 64> : neighbor_count_ring_kernel<NVMK_METRIC_TANIMOTO, false, false, 64>)
                        : (emit ? neighbor_count_ring_kernel<NVMK_METRIC_TANIMOTO, true> : neighbor_count_ring_kernel<NVMK_METRIC_TANIMOTO, false>);
      } else {
        kern = wt == 64 ? (emit ? neighbor_count_ring_kernel<NVMK_METRIC_COSINE, true, false, 64> : neighbor_count_ring_kernel<NVMK_METRIC_COSINE, false, false, 64>)
                        : (emit ? neighbor_count_ring_kernel<NVMK_METRIC_COSINE, true> : neighbor_count_ring_kernel<NVMK_METRIC_COSINE, false>);
      }
      const size_t shmem = static_cast<size_t>(RST) * 2 * RT * RKW * 16 + 2 * 2 * RT * 4 + 2 * RT * 4 + REDGE_CAP * 8 + 16;
      NVMK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(shmem)));
      const bool profile = std::getenv("NVMK_RING_PROFILE") != nullptr && a.metric == NVMK_METRIC_TANIMOTO;
      long long* dProf   = nullptr;
      if (profile) {
        kern = wt == 64 ? (emit ? neighbor_count_ring_kernel<NVMK_METRIC_TANIMOTO, true, true, 64> : neighbor_count_ring_kernel<NVMK_METRIC_TANIMOTO, false, true, 64>)
                        : (emit ? neighbor_count_ring_kernel<NVMK_METRIC_TANIMOTO, true, true> : neighbor_count_ring_kernel<NVMK_METRIC_TANIMOTO, false, true>);
        NVMK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shmem)));
        NVMK_HIP_CHECK(hipMalloc(&dProf, 64 * sizeof(long long)));
      }
      hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(cus)), dim3(wt == 64 ? 1024 : 512), shmem, stream, X.rows, X.popc, a.nX, Y.rows, Y.popc, a.nY, X.L.Wp,
                         at.k1, at.k2, at.adj, a.thr, a.sign, a.symmetric ? 1 : 0, counts, a.edges, a.edgeCursor, a.edgeCapacity, dProf);
      NVMK_LAUNCH_CHECK();
      if (profile) {  // debugging aid: workgroup 0, per wave, 100 MHz ticks
        long long h[64];
        NVMK_HIP_CHECK(hipStreamSynchronize(stream));
        NVMK_HIP_CHECK(hipMemcpy(h, dProf, sizeof(h), hipMemcpyDeviceToHost));
        NVMK_HIP_CHECK(hipFree(dProf));
        for (int w = 0; w < (wt == 64 ? 16 : 8); ++w)
          std::fprintf(stderr, "[ring] wave %d: %lld tiles; per tile: wait+barrier %.2f us, issue+mma %.2f us, epilogue %.2f us\n", w, h[w * 4 + 3],
                       h[w * 4] * 0.01 / h[w * 4 + 3], h[w * 4 + 1] * 0.01 / h[w * 4 + 3], h[w * 4 + 2] * 0.01 / h[w * 4 + 3]);
      }
      return NVMK_OK;
    }
  }
  const int64_t tilesM  = ceil_div<int64_t>(a.nX, TM);
  const int64_t tilesN  = ceil_div<int64_t>(a.nY, TN);
  static const int64_t superE = [] {  // supertile edge for this kernel (NVMK_COUNT_SUPER overrides for experiments)
    const char* e = std::getenv("NVMK_COUNT_SUPER");
    const int   v = e ? std::atoi(e) : 0;
    return static_cast<int64_t>(v > 0 ? v : SUPER);
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
                        unsigned long long, double, double, double, float);
  Kern kern;
  ArithThreshold at = arith_threshold(a.thr, F);
  if (const char* te = std::getenv("NVMK_COUNT_THRESHOLD"); te != nullptr && std::string(te) == "table") at.ok = false;  // tests: force the table form
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
                     (a.bandSkip && a.metric == NVMK_METRIC_TANIMOTO && a.thr > 0.0f && a.xRows == nullptr && a.yRows == nullptr) ? a.thr : 0.0f);
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

}  // namespace fp4
}  // namespace nvmk

extern "C" {

size_t nvmkx_fp4_workspace_bytes(int64_t n, int fp_bits) {
  if (n < 0 || fp_bits <= 0 || fp_bits % 32 != 0) return 0;
  return nvmk::fp4::layout(n, fp_bits).bytes;
}

int nvmkx_fp4_prepare(const uint32_t* d_in, int64_t n, int fp_bits, void* d_workspace, void* stream) {
  return nvmk::fp4::prepare(d_in, nullptr, n, fp_bits, d_workspace, nvmk::as_stream(stream));
}

int nvmkx_cross_similarity_prepared_f64(int metric, const void* d_ws_a, int64_t nA_total, int64_t a_row0, int64_t a_rows,
                                       const void* d_ws_b, int64_t nB, int fp_bits, double* d_out, int64_t ld_out,
                                       void* stream) {
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


// counts[i] += #{ j : sim(x_i, y_j) >= thr } on prepared sets through launch_counts (NVMK_COUNT_KERNEL selects the kernel)
int nvmkx_neighbor_counts_prepared(int metric, const void* d_ws_x, int64_t nX, const void* d_ws_y, int64_t nY, int fp_bits, float thr,
                                   int symmetric, const float* d_table_f, int32_t* d_counts, void* stream) {
  nvmk::fp4::CountArgs a{};
  a.metric    = metric;
  a.thr       = thr;
  a.tableF    = d_table_f;
  a.sign      = 1;
  a.nX        = nX;
  a.nY        = nY;
  a.symmetric = symmetric != 0;
  return nvmk::fp4::launch_counts(a, nvmk::fp4::view(d_ws_x, nX, fp_bits), nvmk::fp4::view(d_ws_y, nY, fp_bits), d_counts,
                                  nvmk::as_stream(stream));
}

}  // extern "C"
