// Dense N x M Tanimoto / cosine cross-similarity on packed fingerprints — gfx950 (MI355X).
//
// Replaces the reference's src/similarity_kernels.cu (tensor-op kernel :96-240, SIMT kernel
// :242-368, launchers :505-582 / :727-799).  The arithmetic is the reference's SIMT statement
// (:350-364): c = popcount(a & b), tanimoto = c / max(1, pa + pb - c) in double, cosine =
// c / sqrt(pa * pb) in double with 0 when c == 0 or the denominator is 0.
//
// Design (not a translation: the reference tiles for 32-wide warps and 1-bit MMA):
//   * one 256-thread workgroup (4 wave64) owns a 128 x 128 output tile; both operand tiles are
//     staged ONCE in LDS for a K chunk of up to 64 words (a whole 2048-bit fingerprint), so
//     there is no barrier inside the popcount loop;
//   * each lane owns an 8 x 8 register tile: per 4-word step it issues 16 ds_read_b128 and
//     512 VALU (v_and_b32 + v_bcnt_u32_b32 with free accumulate) -> LDS runs at ~25 % of its
//     256 B/clk/CU, the VALU is the pole;
//   * LDS rows are stored in 16-byte slots XOR-swizzled by (row >> 2) so that the 16 lanes of
//     every ds_read_b128 lane group hit 16 different slots of the 256-byte bank row;
//   * a lane's 8 columns are two runs of 4 (tx*4.. and 64+tx*4..), so the epilogue writes
//     32 contiguous bytes per lane and 512 contiguous bytes per 16 lanes (full 128-B lines);
//   * workgroup -> tile map walks the A tiles fastest inside groups of 64 tile-rows, so the 8
//     XCD L2s each keep 1/8 of the A group resident while B tiles stream through.
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "options.h"
#include "fp4.h"

namespace nvmk {
namespace sim {

constexpr int TM       = 128;  // output rows per workgroup (rows of A)
constexpr int TN       = 128;  // output cols per workgroup (rows of B)
constexpr int NT       = 256;  // threads per workgroup
constexpr int GROUP_TM = 64;   // tile-rows per scheduling group

// acc += popcount(x): v_bcnt_u32_b32 has a free accumulate operand.  Spelled as inline asm because
// the optimiser otherwise reassociates the adds into bcnt(x, 0) + v_add3 (25 % more VALU).
__device__ __forceinline__ void bcnt_acc(int& acc, const unsigned x) {
  asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x));
}

__device__ __forceinline__ void popc4(const uint4 a, const uint4 b, int& acc) {
  bcnt_acc(acc, a.x & b.x);
  bcnt_acc(acc, a.y & b.y);
  bcnt_acc(acc, a.z & b.z);
  bcnt_acc(acc, a.w & b.w);
}

template <int METRIC> __device__ __forceinline__ double finish(const int c, const int pa, const int pb) {
  if constexpr (METRIC == NVMK_METRIC_TANIMOTO) {
    const int u = pa + pb - c;
    return static_cast<double>(c) / static_cast<double>(u > 1 ? u : 1);
  } else {
    const double denom = sqrt(static_cast<double>(pa) * static_cast<double>(pb));
    return (c == 0 || denom == 0.0) ? 0.0 : static_cast<double>(c) / denom;
  }
}

// Stage `ROWS` rows x S 16-byte slots of one operand into swizzled LDS and accumulate the
// per-row popcounts.  Rows past `n` are clamped (their results are never stored).
template <int S>
__device__ __forceinline__ void stage_tile(const uint4* __restrict__ g,
                                           const int64_t             row0,
                                           const int64_t             n,
                                           const int                 rowStride4,  // uint4 per global row
                                           const int                 chunk,
                                           char*                     lds,
                                           int*                      pc,
                                           const int                 tid) {
  constexpr int PIECES = TM * S;
  constexpr int PASSES = (PIECES + NT - 1) / NT;
#pragma unroll
  for (int t = 0; t < PASSES; ++t) {
    const int  p     = tid + t * NT;
    const bool valid = p < PIECES;
    const int  row   = valid ? p / S : 0;
    const int  slot  = p % S;
    int64_t    grow  = row0 + row;
    grow             = grow < n ? grow : n - 1;
    uint4 v          = g[grow * rowStride4 + chunk * S + slot];
    if (valid) {
      const int sw = slot ^ ((row >> 2) & (S - 1));
      *reinterpret_cast<uint4*>(lds + (row * S + sw) * 16) = v;
    }
    int cnt = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
#pragma unroll
    for (int o = S / 2; o > 0; o >>= 1) {
      cnt += __shfl_xor(cnt, o);
    }
    if (valid && slot == 0) {
      pc[row] = (chunk == 0) ? cnt : pc[row] + cnt;
    }
  }
}

template <int KC, int METRIC>
__global__ __launch_bounds__(NT, 2) void cross_sim_tile_kernel(const uint4* __restrict__ A,
                                                            const int64_t nA,
                                                            const uint4* __restrict__ B,
                                                            const int64_t nB,
                                                            const int     nChunks,
                                                            double* __restrict__ out,
                                                            const int64_t ld,
                                                            const int64_t tilesM,
                                                            const int64_t tilesN,
                                                            const int     vecStore) {
  constexpr int S        = KC / 4;   // 16-byte slots per LDS row
  constexpr int ROWBYTES = S * 16;
  __shared__ __attribute__((aligned(16))) char smem[(TM + TN) * ROWBYTES + (TM + TN) * 4];
  char* sA  = smem;
  char* sB  = smem + TM * ROWBYTES;
  int*  pcA = reinterpret_cast<int*>(smem + (TM + TN) * ROWBYTES);
  int*  pcB = pcA + TM;

  // tile map: blockIdx.y = group of GROUP_TM tile-rows; inside a group tile_m runs fastest
  // (32-bit arithmetic only: a 64-bit divide costs ~300 instructions of prologue).
  const unsigned firstM = blockIdx.y * GROUP_TM;
  const unsigned remM   = static_cast<unsigned>(tilesM) - firstM;
  const unsigned gm     = remM < GROUP_TM ? remM : GROUP_TM;
  const unsigned tile_n = blockIdx.x / gm;
  const unsigned tile_m = firstM + (blockIdx.x - tile_n * gm);
  if (tile_n >= static_cast<unsigned>(tilesN)) {
    return;  // ragged last group: GROUP_TM*tilesN slots launched, only gm*tilesN used
  }

  const int     tid   = threadIdx.x;
  const int     tx    = tid & 15;
  const int     ty    = tid >> 4;
  const int64_t rowA0 = static_cast<int64_t>(tile_m) * TM;
  const int64_t rowB0 = static_cast<int64_t>(tile_n) * TN;
  const int     W4    = nChunks * S;

  int acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[i][j] = 0;
    }
  }

  // swizzled base addresses (see header comment): addr(row, s) = pre(row) ^ (s << 4)
  const int      ra    = ty * 8;
  const unsigned preA0 = static_cast<unsigned>(ra * ROWBYTES) | (static_cast<unsigned>((ra >> 2) & (S - 1)) << 4);
  const unsigned preA1 =
    static_cast<unsigned>((ra + 4) * ROWBYTES) | (static_cast<unsigned>(((ra + 4) >> 2) & (S - 1)) << 4);
  const unsigned preB = static_cast<unsigned>(tx * 4 * ROWBYTES) | (static_cast<unsigned>(tx & (S - 1)) << 4);

  for (int ch = 0; ch < nChunks; ++ch) {
    if (ch > 0) {
      __syncthreads();
    }
    stage_tile<S>(A, rowA0, nA, W4, ch, sA, pcA, tid);
    stage_tile<S>(B, rowB0, nB, W4, ch, sB, pcB, tid);
    __syncthreads();

#pragma clang loop vectorize(disable) unroll(disable)
    for (int s = 0; s < S; ++s) {
      const unsigned sx = static_cast<unsigned>(s) << 4;
      const char*    pa0 = sA + (preA0 ^ sx);
      const char*    pa1 = sA + (preA1 ^ sx);
      const char*    pb  = sB + (preB ^ sx);
      uint4          a[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i]     = *reinterpret_cast<const uint4*>(pa0 + i * ROWBYTES);
        a[i + 4] = *reinterpret_cast<const uint4*>(pa1 + i * ROWBYTES);
      }
      uint4 bv = *reinterpret_cast<const uint4*>(pb);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        uint4 bn = bv;
        if (j < 7) {  // software prefetch of the next B row while this one is consumed
          bn = *reinterpret_cast<const uint4*>(pb + (((j + 1) & 3) + 64 * ((j + 1) >> 2)) * ROWBYTES);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          popc4(a[i], bv, acc[i][j]);
        }
        bv = bn;
      }
    }
  }

  // ---- epilogue: integer counts -> double ratios, 32 contiguous bytes per lane per run ----
  int pbv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    pbv[j] = pcB[tx * 4 + (j & 3) + 64 * (j >> 2)];
  }
  const bool fullTile = (rowA0 + TM <= nA) && (rowB0 + TN <= nB);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t row = rowA0 + ra + i;
    const int     pav = pcA[ra + i];
    if (!fullTile && row >= nA) {
      continue;
    }
    double* orow = out + row * ld;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t col0 = rowB0 + tx * 4 + 64 * h;
      double        v[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        v[jj] = finish<METRIC>(acc[i][h * 4 + jj], pav, pbv[h * 4 + jj]);
      }
      if (fullTile && vecStore) {
        using d2 = __attribute__((ext_vector_type(2))) double;
        d2 lo = {v[0], v[1]};
        d2 hi = {v[2], v[3]};
        __builtin_nontemporal_store(lo, reinterpret_cast<d2*>(orow + col0));
        __builtin_nontemporal_store(hi, reinterpret_cast<d2*>(orow + col0 + 2));
      } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          if (col0 + jj < nB) {
            orow[col0 + jj] = v[jj];
          }
        }
      }
    }
  }
}

// Fallback for widths that are not a multiple of 4 words or for unaligned operands:
// one lane per output element, operands straight from global memory.
template <int METRIC>
__global__ __launch_bounds__(NT) void cross_sim_generic_kernel(const uint32_t* __restrict__ A,
                                                               const int64_t nA,
                                                               const uint32_t* __restrict__ B,
                                                               const int64_t nB,
                                                               const int     W,
                                                               double* __restrict__ out,
                                                               const int64_t ld) {
  const int64_t col = static_cast<int64_t>(blockIdx.x) * 64 + (threadIdx.x & 63);
  const int64_t row = static_cast<int64_t>(blockIdx.y) * 4 + (threadIdx.x >> 6);
  if (row >= nA || col >= nB) {
    return;
  }
  const uint32_t* a = A + row * W;
  const uint32_t* b = B + col * W;
  int c = 0, pa = 0, pb = 0;
  for (int k = 0; k < W; ++k) {
    const uint32_t x = a[k];
    const uint32_t y = b[k];
    c += __popc(x & y);
    pa += __popc(x);
    pb += __popc(y);
  }
  out[row * ld + col] = finish<METRIC>(c, pa, pb);
}

// Path selection.  The matrix-core formulation (similarity_mfma.hip) pays an O(nA + nB) expansion and
// wins by ~3x once the O(nA * nB) term dominates; the VALU popcount kernel below serves small problems
// (e.g. 1 x M "BulkTanimoto" calls) and widths/alignments the FP4 path does not take.
// NVMK_SIM_PATH = auto | valu | mfma overrides (used by the A/B benchmarks and the parity tests).
enum class Path { kAuto, kValu, kMfma };

inline Path path_override() {
  const opt::Text e = opt::get(opt::kSimPath);
  if (e.is("valu")) return Path::kValu;
  if (e.is("mfma")) return Path::kMfma;
  return Path::kAuto;
}

template <int METRIC>
int launch_mfma(const uint32_t* a, int64_t nA, const uint32_t* b, int64_t nB, int fpBits, double* out, int64_t ld,
                hipStream_t stream) {
  const bool          same = (a == b && nA == nB);
  const fp4::Layout   LA   = fp4::layout(nA, fpBits);
  const fp4::Layout   LB   = fp4::layout(nB, fpBits);
  StreamScratch       wsA, wsB;
  NVMK_HIP_CHECK(wsA.alloc(LA.bytes, stream));
  int rc = fp4::prepare(a, nullptr, nA, fpBits, wsA.ptr, stream);
  if (rc != NVMK_OK) return rc;
  if (!same) {
    NVMK_HIP_CHECK(wsB.alloc(LB.bytes, stream));
    rc = fp4::prepare(b, nullptr, nB, fpBits, wsB.ptr, stream);
    if (rc != NVMK_OK) return rc;
  }
  return fp4::launch_dense(METRIC, fp4::view(wsA.ptr, nA, fpBits), fp4::view(same ? wsA.ptr : wsB.ptr, nB, fpBits), out,
                           ld, stream);
}

template <int METRIC>
int launch(const uint32_t* a, int64_t nA, const uint32_t* b, int64_t nB, int fpBits, double* out, int64_t ld,
           hipStream_t stream) {
  NVMK_REQUIRE(fpBits > 0 && fpBits % 32 == 0, "cross similarity: fp_bits must be a positive multiple of 32, got %d",
               fpBits);
  NVMK_REQUIRE(nA >= 0 && nB >= 0, "cross similarity: negative row count (%lld, %lld)", (long long)nA, (long long)nB);
  if (nA == 0 || nB == 0) {
    return NVMK_OK;
  }
  NVMK_REQUIRE(a != nullptr && b != nullptr && out != nullptr, "cross similarity: NULL buffer");
  NVMK_REQUIRE(ld >= nB, "cross similarity: ld_out (%lld) < nB (%lld)", (long long)ld, (long long)nB);
  const int  W       = fpBits / 32;
  const bool aligned = (reinterpret_cast<uintptr_t>(a) % 16 == 0) && (reinterpret_cast<uintptr_t>(b) % 16 == 0);
  {
    const Path   p       = path_override();
    const double pairs   = static_cast<double>(nA) * static_cast<double>(nB);
    const bool   worthIt = pairs >= 4.0e6 && nA >= 64 && nB >= 64 && W >= 4;
    if (p == Path::kMfma || (p == Path::kAuto && worthIt)) {
      return launch_mfma<METRIC>(a, nA, b, nB, fpBits, out, ld, stream);
    }
  }
  if (W % 4 != 0 || !aligned) {
    const dim3 grid(static_cast<unsigned>(ceil_div<int64_t>(nB, 64)), static_cast<unsigned>(ceil_div<int64_t>(nA, 4)));
    NVMK_REQUIRE(ceil_div<int64_t>(nA, 4) <= 65535 * 1024LL, "cross similarity: generic path supports at most %lld rows",
                 65535LL * 4096);
    hipLaunchKernelGGL(cross_sim_generic_kernel<METRIC>, grid, dim3(NT), 0, stream, a, nA, b, nB, W, out, ld);
    NVMK_LAUNCH_CHECK();
    return NVMK_OK;
  }
  const int64_t tilesM   = ceil_div<int64_t>(nA, TM);
  const int64_t tilesN   = ceil_div<int64_t>(nB, TN);
  const int64_t groups   = ceil_div<int64_t>(tilesM, GROUP_TM);
  NVMK_REQUIRE(groups <= 65535 && GROUP_TM * tilesN <= 0x7fffffffLL,
               "cross similarity: problem too large for one launch (%lld x %lld tiles)", (long long)tilesM,
               (long long)tilesN);
  const int   vecStore = (ld % 2 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0);
  const auto* a4       = reinterpret_cast<const uint4*>(a);
  const auto* b4       = reinterpret_cast<const uint4*>(b);
  const dim3  grid(static_cast<unsigned>(GROUP_TM * tilesN), static_cast<unsigned>(groups));
#define NVMK_SIM_LAUNCH(KC)                                                                                     \
  hipLaunchKernelGGL((cross_sim_tile_kernel<KC, METRIC>), grid, dim3(NT), 0, stream, a4, nA, b4, nB, W / (KC), out, \
                     ld, tilesM, tilesN, vecStore)
  if (W % 64 == 0) {
    NVMK_SIM_LAUNCH(64);
  } else if (W % 32 == 0) {
    NVMK_SIM_LAUNCH(32);
  } else if (W % 16 == 0) {
    NVMK_SIM_LAUNCH(16);
  } else if (W % 8 == 0) {
    NVMK_SIM_LAUNCH(8);
  } else {
    NVMK_SIM_LAUNCH(4);
  }
#undef NVMK_SIM_LAUNCH
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

}  // namespace sim
}  // namespace nvmk

extern "C" {

int nvmk_cross_tanimoto_f64(const uint32_t* d_a, int64_t nA, const uint32_t* d_b, int64_t nB, int fp_bits,
                            double* d_out, int64_t ld_out, void* stream) {
  NVMK_MARK_ENTRY();
  return nvmk::sim::launch<NVMK_METRIC_TANIMOTO>(d_a, nA, d_b, nB, fp_bits, d_out, ld_out, nvmk::as_stream(stream));
}

int nvmk_cross_cosine_f64(const uint32_t* d_a, int64_t nA, const uint32_t* d_b, int64_t nB, int fp_bits, double* d_out,
                          int64_t ld_out, void* stream) {
  NVMK_MARK_ENTRY();
  return nvmk::sim::launch<NVMK_METRIC_COSINE>(d_a, nA, d_b, nB, fp_bits, d_out, ld_out, nvmk::as_stream(stream));
}

}  // extern "C"
