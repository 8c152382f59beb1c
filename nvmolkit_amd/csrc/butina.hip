// Matrix-free neighbour counting, fused Taylor-Butina and dense-matrix Taylor-Butina — gfx950.
//
// Replaces (reference paths):
//   nvmolkit/_fusedButina.py:99-179   _update_neighbor_count_kernel (Triton + PTX popc)
//   nvmolkit/_fusedButina.py:182-246  _extract_cluster_singleton_kernel
//   nvmolkit/clustering.py:99-189     fused_butina (Python loop, one host sync per cluster)
//   src/butina.cu:913-1071            butinaGpu (dense matrix, CUDA-graph WHILE loops)
//
// Design notes (MI355X-first, not a translation):
//   * neighbour counting reuses the dense kernel's 128x128 LDS tile / 8x8 register tile, but a
//     workgroup keeps its A tile resident and walks a STRIP of B tiles, prefetching the next B tile
//     into registers while the popcount loop runs; row counts live in registers for the whole strip
//     and are flushed with one atomic per row per strip.
//   * the reference's per-pair float32 division is replaced by a table: for Tanimoto the predicate
//     float(c)/float(pa+pb-c) >= thr is monotone in c for fixed s = pa+pb, so tmin[s] = the smallest
//     such c is built once per call (with real float32 divisions, so the predicate is bit-identical)
//     and the per-pair work is one LDS lookup and one compare.
//   * the fused Butina loop runs on the device: argmax / extract+compact / subtract kernels are
//     enqueued in batches and the host reads one status word per batch instead of syncing per cluster.
//   * the fingerprint matrix is never compacted; kernels gather rows through index lists.
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include <cstdlib>
#include <cstring>

#include <hipcub/hipcub.hpp>

#include "common.h"
#include "options.h"
#include "fp4.h"

namespace nvmk {
namespace butina {

constexpr int      TM       = 128;
constexpr int      TN       = 128;
constexpr int      NT       = 256;
constexpr int      STRIP    = 32;      // B tiles walked by one workgroup
constexpr uint16_t NEVER    = 0xFFFF;  // table sentinel: no c satisfies the predicate

__device__ __forceinline__ void bcnt_acc(int& acc, const unsigned x) {
  asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x));
}

__device__ __forceinline__ void popc4(const uint4 a, const uint4 b, int& acc) {
  bcnt_acc(acc, a.x & b.x);
  bcnt_acc(acc, a.y & b.y);
  bcnt_acc(acc, a.z & b.z);
  bcnt_acc(acc, a.w & b.w);
}

__device__ __forceinline__ int popc_u4(const uint4 v) { return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }

// float32 neighbour predicate, exactly the oracle's / reference's (nvmolkit/_fusedButina.py:160-173).
template <int METRIC> __device__ __forceinline__ bool is_neighbor(const int c, const int pa, const int pb, const float thr) {
  if constexpr (METRIC == NVMK_METRIC_TANIMOTO) {
    const int u = pa + pb - c;
    if (u <= 0) return false;
    return static_cast<float>(c) / static_cast<float>(u) >= thr;
  } else {
    const float denom = sqrtf(static_cast<float>(pa) * static_cast<float>(pb));
    if (!(denom > 0.0f)) return false;
    return static_cast<float>(c) / denom >= thr;
  }
}

// tmin[s], s = pa + pb in [0, 2*F]: smallest c with float(c)/float(s-c) >= thr (NEVER if none);
// entries (2F, 4F+2] are NEVER so that a sentinel popcount of 2F+1 disables a padded row or column.
// tableF holds the same thresholds as floats (+inf for NEVER): the matrix-core kernel compares its f32 accumulators
// (exact integers) against it without a conversion.
__global__ void build_tanimoto_table_kernel(uint16_t* __restrict__ table, float* __restrict__ tableF, const int F,
                                            const float thr) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > 4 * F + 2) return;
  uint16_t v = NEVER;
  if (s >= 1 && s <= 2 * F) {
    // predicate is monotone non-decreasing in c on [0, s-1]
    int lo = 0, hi = s;  // answer in [lo, hi], hi == s means none
    while (lo < hi) {
      const int  mid = (lo + hi) >> 1;
      const bool ok  = static_cast<float>(mid) / static_cast<float>(s - mid) >= thr;
      if (ok) {
        hi = mid;
      } else {
        lo = mid + 1;
      }
    }
    if (lo < s) v = static_cast<uint16_t>(lo);
  }
  table[s]  = v;
  tableF[s] = (v == NEVER) ? __builtin_inff() : static_cast<float>(v);
}

// ---- staging helpers -------------------------------------------------------------------------

template <int S> struct Geo {
  static constexpr int SWZ      = (S < 16 ? S : 16) - 1;  // swizzle mask on the low slot bits
  static constexpr int ROWBYTES = S * 16;
  static constexpr int PIECES   = TM * S;
  static constexpr int PASSES   = (PIECES + NT - 1) / NT;
};

// Load this thread's pieces of a 128-row tile into registers.  Rows are gathered through `rows`
// (NULL = identity); rows past n are clamped to row n-1 (never stored / counted).
template <int S>
__device__ __forceinline__ void tile_fetch(uint4 (&v)[Geo<S>::PASSES], const uint4* __restrict__ g,
                                           const int32_t* __restrict__ rows, const int64_t row0, const int64_t n,
                                           const int tid) {
#pragma unroll
  for (int t = 0; t < Geo<S>::PASSES; ++t) {
    const int p    = tid + t * NT;
    const int row  = (p < Geo<S>::PIECES) ? p / S : 0;
    const int slot = p % S;
    int64_t   r    = row0 + row;
    r              = r < n ? r : n - 1;
    const int64_t phys = rows ? static_cast<int64_t>(rows[r]) : r;
    v[t]               = g[phys * S + slot];
  }
}

// Write fetched pieces to swizzled LDS; per-row popcounts go to pc[] (rows >= valid rows get `padPc`).
template <int S>
__device__ __forceinline__ void tile_commit(const uint4 (&v)[Geo<S>::PASSES], char* lds, int* pc, const int64_t row0,
                                            const int64_t n, const int padPc, const int tid) {
#pragma unroll
  for (int t = 0; t < Geo<S>::PASSES; ++t) {
    const int  p     = tid + t * NT;
    const bool valid = p < Geo<S>::PIECES;
    const int  row   = valid ? p / S : 0;
    const int  slot  = p % S;
    if (valid) {
      const int sw = slot ^ ((row >> 2) & Geo<S>::SWZ);
      *reinterpret_cast<uint4*>(lds + (row * S + sw) * 16) = v[t];
    }
    int cnt = popc_u4(v[t]);
#pragma unroll
    for (int o = (S < 64 ? S : 64) / 2; o > 0; o >>= 1) {
      cnt += __shfl_xor(cnt, o);
    }
    if (valid && slot == 0) {
      pc[row] = (row0 + row < n) ? cnt : padPc;
    }
  }
}

// counts[xrow] += sign * #{ y rows that are neighbours of x row }.
// grid = (strips, tilesM): blockIdx.y = A tile, blockIdx.x = strip of STRIP B tiles.
// nXdev / nYdev: optional device-side row counts (the Butina loop launches with host-side upper
// bounds and lets the kernel read the true sizes); NULL = use nX / nY.
template <int S, int METRIC>
__global__ __launch_bounds__(NT, 2) void neighbor_count_kernel(const uint4* __restrict__ X,
                                                               const int32_t* __restrict__ xRows,
                                                               int64_t       nX,
                                                               const int32_t* __restrict__ nXdev,
                                                               const uint4* __restrict__ Y,
                                                               const int32_t* __restrict__ yRows,
                                                               int64_t       nY,
                                                               const int32_t* __restrict__ nYdev,
                                                               const uint16_t* __restrict__ table,
                                                               const int     F,
                                                               const float   thr,
                                                               const int     sign,
                                                               int32_t* __restrict__ counts) {
  using G                = Geo<S>;
  constexpr int ROWBYTES = G::ROWBYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char*     sA   = smem;
  char*     sB   = smem + TM * ROWBYTES;
  int*      pcA  = reinterpret_cast<int*>(smem + (TM + TN) * ROWBYTES);
  int*      pcB  = pcA + TM;
  uint16_t* sTab = reinterpret_cast<uint16_t*>(pcB + TN);

  if (nXdev) nX = *nXdev;
  if (nYdev) nY = *nYdev;
  const int64_t rowA0  = static_cast<int64_t>(blockIdx.y) * TM;
  const int64_t tilesN = (nY + TN - 1) / TN;
  const int64_t jt0    = static_cast<int64_t>(blockIdx.x) * STRIP;
  if (rowA0 >= nX || jt0 >= tilesN) {
    return;
  }
  const int64_t jt1 = (jt0 + STRIP < tilesN) ? jt0 + STRIP : tilesN;

  const int tid = threadIdx.x;
  const int tx  = tid & 15;
  const int ty  = tid >> 4;
  const int ra  = ty * 8;

  uint4 pre[G::PASSES];
  tile_fetch<S>(pre, X, xRows, rowA0, nX, tid);
  tile_commit<S>(pre, sA, pcA, rowA0, nX, 0, tid);
  if constexpr (METRIC == NVMK_METRIC_TANIMOTO) {
    for (int k = tid; k <= 3 * F + 1; k += NT) {
      sTab[k] = table[k];
    }
  }
  tile_fetch<S>(pre, Y, yRows, jt0 * TN, nY, tid);

  const unsigned preA0 = static_cast<unsigned>(ra * ROWBYTES) | (static_cast<unsigned>((ra >> 2) & G::SWZ) << 4);
  const unsigned preA1 =
    static_cast<unsigned>((ra + 4) * ROWBYTES) | (static_cast<unsigned>(((ra + 4) >> 2) & G::SWZ) << 4);
  const unsigned preB = static_cast<unsigned>(tx * 4 * ROWBYTES) | (static_cast<unsigned>(tx & G::SWZ) << 4);

  int rowcnt[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) rowcnt[i] = 0;
  const int padPc = (METRIC == NVMK_METRIC_TANIMOTO) ? 2 * F + 1 : 0;

  for (int64_t jt = jt0; jt < jt1; ++jt) {
    __syncthreads();  // previous tile fully consumed (and, first time, sA / sTab visible after the next barrier)
    tile_commit<S>(pre, sB, pcB, jt * TN, nY, padPc, tid);
    __syncthreads();
    if (jt + 1 < jt1) {
      tile_fetch<S>(pre, Y, yRows, (jt + 1) * TN, nY, tid);  // in flight during the popcount loop
    }

    int acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = 0;
    }
#pragma clang loop vectorize(disable) unroll(disable)
    for (int s = 0; s < S; ++s) {
      // slot s of a row lives at (s & ~SWZ) | ((s ^ g(row)) & SWZ): XOR only touches the low bits
      const unsigned sx  = static_cast<unsigned>(s) << 4;
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
        if (j < 7) {
          bn = *reinterpret_cast<const uint4*>(pb + (((j + 1) & 3) + 64 * ((j + 1) >> 2)) * ROWBYTES);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          popc4(a[i], bv, acc[i][j]);
        }
        bv = bn;
      }
    }

    // threshold the 8x8 counts
    int pbv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) pbv[j] = pcB[tx * 4 + (j & 3) + 64 * (j >> 2)];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int pav = pcA[ra + i];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (METRIC == NVMK_METRIC_TANIMOTO) {
          const int t = sTab[pav + pbv[j]];
          rowcnt[i] += (acc[i][j] >= t) ? 1 : 0;
        } else {
          rowcnt[i] += is_neighbor<METRIC>(acc[i][j], pav, pbv[j], thr) ? 1 : 0;
        }
      }
    }
  }

  // flush: reduce over the 16 lanes that share a row, one atomic per row per strip
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int v = rowcnt[i];
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    const int64_t r = rowA0 + ra + i;
    if (tx == 0 && r < nX && v != 0) {
      const int64_t phys = xRows ? static_cast<int64_t>(xRows[r]) : r;
      atomicAdd(&counts[phys], sign * v);
    }
  }
}

// Fallback for widths without an LDS-tiled instantiation: one wave per x row, lanes stride y rows.
template <int METRIC>
__global__ __launch_bounds__(NT) void neighbor_count_generic_kernel(const uint32_t* __restrict__ X,
                                                                    const int32_t* __restrict__ xRows,
                                                                    int64_t       nX,
                                                                    const int32_t* __restrict__ nXdev,
                                                                    const uint32_t* __restrict__ Y,
                                                                    const int32_t* __restrict__ yRows,
                                                                    int64_t       nY,
                                                                    const int32_t* __restrict__ nYdev,
                                                                    const int     W,
                                                                    const float   thr,
                                                                    const int     sign,
                                                                    int32_t* __restrict__ counts) {
  if (nXdev) nX = *nXdev;
  if (nYdev) nY = *nYdev;
  const int     lane = threadIdx.x & 63;
  const int64_t r    = static_cast<int64_t>(blockIdx.x) * (NT / 64) + (threadIdx.x >> 6);
  if (r >= nX) return;
  const int64_t   px = xRows ? static_cast<int64_t>(xRows[r]) : r;
  const uint32_t* x  = X + px * W;
  int             pa = 0;
  for (int k = 0; k < W; ++k) pa += __popc(x[k]);
  int n = 0;
  for (int64_t j = lane; j < nY; j += 64) {
    const int64_t   py = yRows ? static_cast<int64_t>(yRows[j]) : j;
    const uint32_t* y  = Y + py * W;
    int             c = 0, pb = 0;
    for (int k = 0; k < W; ++k) {
      c += __popc(x[k] & y[k]);
      pb += __popc(y[k]);
    }
    n += is_neighbor<METRIC>(c, pa, pb, thr) ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
  if (lane == 0 && n != 0) atomicAdd(&counts[px], sign * n);
}

inline bool force_valu() { return opt::get(opt::kSimPath).is("valu"); }
inline bool force_mfma() { return opt::get(opt::kSimPath).is("mfma"); }

// NVMK_BUTINA_ROUNDS=dense keeps the round loop that streams the fingerprint matrix (tests run both formulations)
inline bool dense_rounds() { return opt::get(opt::kButinaRounds).is("dense"); }
// NVMK_BUTINA_ROUNDS=serial: the one-workgroup round loop (sparse_loop_kernel) instead of the parallel rounds
inline bool serial_rounds() { return opt::get(opt::kButinaRounds).is("serial"); }

struct CountPlan {
  int             metric;
  int             fpBits;
  float           thr;
  const uint16_t* table;  // device, Tanimoto only
  const float*    tableF; // the same thresholds as floats (+inf = never)
};

template <int METRIC>
int launch_counts_t(const CountPlan& plan, const uint32_t* x, const int32_t* xRows, int64_t nX, const int32_t* nXdev,
                    const uint32_t* y, const int32_t* yRows, int64_t nY, const int32_t* nYdev, int sign,
                    int32_t* counts, hipStream_t stream) {
  if (nX <= 0 || nY <= 0) return NVMK_OK;
  const int  W       = plan.fpBits / 32;
  const bool aligned = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(y) % 16 == 0);
  const int  S       = W / 4;
  const bool tiled   = aligned && (W % 4 == 0) && (S == 1 || S == 2 || S == 4 || S == 8 || S == 16 || S == 32);
  if (!tiled) {
    const int64_t blocks = ceil_div<int64_t>(nX, NT / 64);
    NVMK_REQUIRE(blocks <= 0x7fffffffLL, "neighbor counts: too many rows for the generic path");
    hipLaunchKernelGGL(neighbor_count_generic_kernel<METRIC>, dim3(static_cast<unsigned>(blocks)), dim3(NT), 0, stream,
                       x, xRows, nX, nXdev, y, yRows, nY, nYdev, W, plan.thr, sign, counts);
    NVMK_LAUNCH_CHECK();
    return NVMK_OK;
  }
  const int64_t tilesM = ceil_div<int64_t>(nX, TM);
  const int64_t strips = ceil_div<int64_t>(ceil_div<int64_t>(nY, TN), STRIP);
  NVMK_REQUIRE(tilesM <= 65535 && strips <= 0x7fffffffLL, "neighbor counts: too many rows for one launch (%lld)",
               (long long)nX);
  const dim3   grid(static_cast<unsigned>(strips), static_cast<unsigned>(tilesM));
  const size_t tabBytes = (METRIC == NVMK_METRIC_TANIMOTO) ? (static_cast<size_t>(3 * plan.fpBits + 2) * 2 + 15) / 16 * 16 : 0;
  const size_t shmem    = static_cast<size_t>(TM + TN) * S * 16 + (TM + TN) * 4 + tabBytes;
  const auto*  x4       = reinterpret_cast<const uint4*>(x);
  const auto*  y4       = reinterpret_cast<const uint4*>(y);
#define NVMK_NC_LAUNCH(SS)                                                                                          \
  hipLaunchKernelGGL((neighbor_count_kernel<SS, METRIC>), grid, dim3(NT), shmem, stream, x4, xRows, nX, nXdev, y4,  \
                     yRows, nY, nYdev, plan.table, plan.fpBits, plan.thr, sign, counts)
  switch (S) {
    case 1: NVMK_NC_LAUNCH(1); break;
    case 2: NVMK_NC_LAUNCH(2); break;
    case 4: NVMK_NC_LAUNCH(4); break;
    case 8: NVMK_NC_LAUNCH(8); break;
    case 16: NVMK_NC_LAUNCH(16); break;
    default: NVMK_NC_LAUNCH(32); break;
  }
#undef NVMK_NC_LAUNCH
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

int launch_counts(const CountPlan& plan, const uint32_t* x, const int32_t* xRows, int64_t nX, const int32_t* nXdev,
                  const uint32_t* y, const int32_t* yRows, int64_t nY, const int32_t* nYdev, int sign, int32_t* counts,
                  hipStream_t stream) {
  return plan.metric == NVMK_METRIC_TANIMOTO ?
           launch_counts_t<NVMK_METRIC_TANIMOTO>(plan, x, xRows, nX, nXdev, y, yRows, nY, nYdev, sign, counts, stream) :
           launch_counts_t<NVMK_METRIC_COSINE>(plan, x, xRows, nX, nXdev, y, yRows, nY, nYdev, sign, counts, stream);
}

int make_plan(CountPlan& plan, StreamScratch& tableMem, int metric, int fpBits, float thr, hipStream_t stream) {
  NVMK_REQUIRE(metric == NVMK_METRIC_TANIMOTO || metric == NVMK_METRIC_COSINE, "unknown metric %d", metric);
  NVMK_REQUIRE(fpBits > 0 && fpBits % 32 == 0, "fp_bits must be a positive multiple of 32, got %d", fpBits);
  NVMK_REQUIRE(fpBits <= 16384, "fp_bits > 16384 is not supported by the neighbour kernels (got %d)", fpBits);
  plan.metric = metric;
  plan.fpBits = fpBits;
  plan.thr    = thr;
  plan.table  = nullptr;
  plan.tableF = nullptr;
  if (metric == NVMK_METRIC_TANIMOTO) {
    const int    entries = 4 * fpBits + 3;
    const size_t u16Bytes = (static_cast<size_t>(entries) * sizeof(uint16_t) + 255) / 256 * 256;
    NVMK_HIP_CHECK(tableMem.alloc(u16Bytes + static_cast<size_t>(entries) * sizeof(float), stream));
    auto* tf = reinterpret_cast<float*>(tableMem.as<char>() + u16Bytes);
    hipLaunchKernelGGL(build_tanimoto_table_kernel, dim3(ceil_div(entries, 256)), dim3(256), 0, stream,
                       tableMem.as<uint16_t>(), tf, fpBits, thr);
    NVMK_LAUNCH_CHECK();
    plan.table  = tableMem.as<uint16_t>();
    plan.tableF = tf;
  }
  return NVMK_OK;
}

// ===================== fused Butina: device-resident round loop ================================

// Device-side loop state (one struct in global memory, zeroed before use).
struct LoopState {
  unsigned long long bestKey[2];  // ((u64)count << 32) | row, ping-pong by round parity
  int32_t            nAlive;      // live rows in the current alive list
  int32_t            nRemoved;    // members of the cluster extracted this round (size of `removed`)
  int32_t            front;       // next free slot at the front of clusterIndices (greedy clusters)
  int32_t            back;        // next free slot at the back (harvested singletons), counts down
  int32_t            nClusters;   // greedy clusters written so far
  int32_t            done;        // set when the max degree reaches 0 or nothing is alive
  int32_t            lastMax;     // degree of the most recent centroid (upper bound for later rounds)
  int32_t            finalParity; // which alive list holds the degree-0 leftovers when done
  // sparse round loop: the bucket of rows that hold the current maximal degree
  int32_t            bucketMax;   // scratch of bucket_max_kernel (reset by the loop kernel)
  int32_t            curDegree;   // degree of the bucket in `cand`
  int32_t            nCand;       // rows in `cand` (descending row order)
  int32_t            parity;      // which of L0 / L1 is the harvest list of the next round
  // parallel rounds (par_* kernels): undecided-candidate lists and where the bucket's clusters are emitted
  int32_t            nU[2];       // entries of the two undecided lists
  int32_t            emitFront;   // clusterIndices position / cluster number of the bucket's first cluster
  int32_t            emitCluster;
  int32_t            nSel;        // clusters the bucket formed
};
static_assert(sizeof(LoopState) <= 32 * sizeof(int32_t), "LoopState must fit the 32-word state block");

__global__ void init_state_kernel(LoopState* __restrict__ st, const int32_t n) {
  st->bestKey[0]  = 0ull;
  st->bestKey[1]  = 0ull;
  st->nAlive      = n;
  st->nRemoved    = 0;
  st->front       = 0;
  st->back        = n - 1;
  st->nClusters   = 0;
  st->done        = 0;
  st->lastMax     = n;
  st->finalParity = 0;
  st->bucketMax   = 0;
  st->curDegree   = 0;
  st->nCand       = 0;
  st->parity      = 0;
  st->nU[0] = st->nU[1] = 0;
  st->emitFront = st->emitCluster = st->nSel = 0;
}

__global__ void iota_kernel(int32_t* __restrict__ rows, const int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) rows[i] = static_cast<int32_t>(i);
}

// argmax over the alive list with ties toward the HIGHEST row (clustering.py:159).
// Also resets the other parity's key for the next round.
__global__ __launch_bounds__(NT) void argmax_kernel(LoopState* __restrict__ st, const int32_t* __restrict__ alive,
                                                    const int32_t* __restrict__ counts, const int parity) {
  if (st->done) return;
  const int n = st->nAlive;
  if (blockIdx.x == 0 && threadIdx.x == 0) st->bestKey[parity ^ 1] = 0ull;
  unsigned long long best = 0ull;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * NT + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * NT) {
    const int32_t r = alive[i];
    const int32_t c = counts[r];
    if (c > 0) {
      const unsigned long long key = (static_cast<unsigned long long>(c) << 32) | static_cast<unsigned>(r);
      best                         = key > best ? key : best;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(best, o);
    best                           = other > best ? other : best;
  }
  __shared__ unsigned long long wbest[NT / 64];
  if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < NT / 64; ++w) best = wbest[w] > best ? wbest[w] : best;
    if (best != 0ull) atomicMax(&st->bestKey[parity], best);
  }
}

// One round of extraction (nvmolkit/_fusedButina.py:182-246) fused with alive-list compaction:
//   neighbours of the centroid -> cluster (front of clusterIndices) and the `removed` list;
//   other rows with degree 1   -> singleton tail (back of clusterIndices);
//   everything else            -> next alive list.
// 16 lanes cooperate on one row (16 B each for 2048-bit fingerprints, looping for wider ones).  Each
// workgroup owns a contiguous slice of the alive list, collects its survivors in LDS and reserves its
// output range with ONE global atomic (a per-row atomic on one counter serialises at ~88 atomics/us and
// made this kernel 80 % of the Butina time at 300k rows).
constexpr int EXTRACT_ROWS = 1024;  // alive rows per workgroup

template <int METRIC>
__global__ __launch_bounds__(NT) void extract_kernel(LoopState* __restrict__ st, const uint4* __restrict__ X, const int W4,
                                                     const int32_t* __restrict__ aliveIn, int32_t* __restrict__ aliveOut,
                                                     int32_t* __restrict__ nAliveOut, int32_t* __restrict__ removed,
                                                     const int32_t* __restrict__ counts,
                                                     int32_t* __restrict__ clusterIndices, const float thr,
                                                     const int parity) {
  if (st->done) return;
  const unsigned long long key = st->bestKey[parity];
  if (key == 0ull) return;  // handled by finish_round_kernel
  const int centroid = static_cast<int>(key & 0xffffffffull);
  const int n        = st->nAlive;
  const int first    = blockIdx.x * EXTRACT_ROWS;
  if (first >= n) return;
  const int last = (first + EXTRACT_ROWS < n) ? first + EXTRACT_ROWS : n;

  __shared__ int32_t keep[EXTRACT_ROWS];
  __shared__ int     nKeep;
  __shared__ int     base;
  if (threadIdx.x == 0) nKeep = 0;
  __syncthreads();

  const int sub = threadIdx.x & 15;
  for (int i0 = first; i0 < last; i0 += NT / 16) {
    const int     i     = i0 + (threadIdx.x >> 4);
    const bool    valid = i < last;
    const int32_t r     = valid ? aliveIn[i] : centroid;
    int           c = 0, pa = 0, pb = 0;
    for (int k = sub; k < W4; k += 16) {
      const uint4 cv = X[static_cast<int64_t>(centroid) * W4 + k];
      const uint4 rv = X[static_cast<int64_t>(r) * W4 + k];
      pa += popc_u4(cv);
      pb += popc_u4(rv);
      c += __popc(cv.x & rv.x) + __popc(cv.y & rv.y) + __popc(cv.z & rv.z) + __popc(cv.w & rv.w);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      c += __shfl_xor(c, o);
      pa += __shfl_xor(pa, o);
      pb += __shfl_xor(pb, o);
    }
    if (valid && sub == 0) {
      const bool nb = (r == centroid) || is_neighbor<METRIC>(c, pa, pb, thr);
      if (nb) {
        clusterIndices[atomicAdd(&st->front, 1)] = r;
        removed[atomicAdd(&st->nRemoved, 1)]     = r;
      } else if (counts[r] == 1) {
        clusterIndices[atomicSub(&st->back, 1)] = r;
      } else {
        keep[atomicAdd(&nKeep, 1)] = r;  // LDS atomic
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) base = atomicAdd(nAliveOut, nKeep);
  __syncthreads();
  for (int k = threadIdx.x; k < nKeep; k += NT) aliveOut[base + k] = keep[k];
}

// Bookkeeping between extract and subtract (single thread): record the cluster, roll the alive
// count, detect termination.
__global__ void finish_round_kernel(LoopState* __restrict__ st, int32_t* __restrict__ nAliveNext,
                                    int32_t* __restrict__ offsets, int32_t* __restrict__ centroids, const int parity) {
  if (st->done) return;
  const unsigned long long key = st->bestKey[parity];
  if (key == 0ull) {  // max degree 0 (or nothing alive): stop; leftovers sit in this round's INPUT list
    st->done        = 1;
    st->finalParity = parity;
    return;
  }
  st->lastMax       = static_cast<int32_t>(key >> 32);
  const int k       = st->nClusters;
  centroids[k]      = static_cast<int32_t>(key & 0xffffffffull);
  offsets[k + 1]    = st->front;
  st->nClusters     = k + 1;
  st->nAlive        = *nAliveNext;
  if (st->nAlive == 0) {
    st->done        = 1;
    st->finalParity = parity ^ 1;
  }
}

// After subtract: reset the per-round counters for the next round.
__global__ void reset_round_kernel(LoopState* __restrict__ st, int32_t* __restrict__ nAliveNextOfNextRound) {
  st->nRemoved             = 0;
  *nAliveNextOfNextRound   = 0;
}

// ===================== fused Butina on the sparse neighbour graph =================================
// At a useful threshold the neighbour graph is sparse (mean degree ~50 at N = 1M): the first all-pairs pass also
// emits every neighbour pair once (matrix-core kernel, EMIT), the pairs become a CSR adjacency, and a round then
// touches only the centroid's list and its members' lists instead of streaming the whole fingerprint matrix twice
// (extract: packed rows, subtract: FP4 rows — 1.3 GB per round at N = 1M, 223 us per round measured).
// Semantics are the oracle's (oracle_similarity.c orc_butina_fused): centroid = LAST row with the maximal degree;
// rows whose degree is 1 in a round that picks another centroid are harvested as singletons — a row's degree drops
// to 1 only in some round's subtract, so the harvest of round k + 1 is exactly the list L collected in round k
// (L0 = rows that start at degree 1); when no row of degree >= 2 is left, the last (highest) row of L becomes a
// greedy cluster of its own and the rest are singletons, as in the dense-round formulation.

__global__ void edge_degree_kernel(const int2* __restrict__ edges, const unsigned long long nEdges,
                                   unsigned long long* __restrict__ deg) {
  for (unsigned long long e = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < nEdges;
       e += static_cast<unsigned long long>(gridDim.x) * blockDim.x) {
    const int2 p = edges[e];
    atomicAdd(&deg[p.x], 1ull);
    atomicAdd(&deg[p.y], 1ull);
  }
}

__global__ void csr_fill_kernel(const int2* __restrict__ edges, const unsigned long long nEdges,
                                const unsigned long long* __restrict__ offsets, unsigned int* __restrict__ cursor,
                                int32_t* __restrict__ nbr) {
  for (unsigned long long e = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < nEdges;
       e += static_cast<unsigned long long>(gridDim.x) * blockDim.x) {
    const int2 p                                 = edges[e];
    nbr[offsets[p.x] + atomicAdd(&cursor[p.x], 1u)] = p.y;
    nbr[offsets[p.y] + atomicAdd(&cursor[p.y], 1u)] = p.x;
  }
}

// L0 = rows whose degree (self included) is exactly 1
__global__ void sparse_init_kernel(const int32_t n, const int32_t* __restrict__ counts, int32_t* __restrict__ L0,
                                   int32_t* __restrict__ nL0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (counts[i] == 1) L0[atomicAdd(nL0, 1)] = i;
}

// ---- the round loop on the device ------------------------------------------------------------------
// A round needs the LAST row with the maximal degree (clustering.py:159).  Degrees only ever decrease, so the rows that
// hold the current maximum D form a bucket that can only shrink: it is compacted once, in DESCENDING row order, and a
// persistent one-workgroup kernel then runs round after round off it — centroid = first bucket entry that is still
// alive with degree D — with no launch, no 1M-row argmax and no host sync per round (those cost 24.5 us x 20 032
// rounds = 0.49 of the 1.0 s at N = 1M).  When the bucket runs dry the kernel returns and the next epoch (four small
// multi-workgroup kernels: max, count, scan, fill) builds the bucket of the next lower degree present; the number of
// epochs is bounded by the number of distinct degrees (80 at N = 1M).
// Inside the persistent kernel, data written by the workgroup itself (atomics execute in L2, stores write through)
// is re-read with agent-scope loads, which bypass the CU's L1: a plain load could hit a stale line.

constexpr int BUCKET_ROWS = 1024;  // rows per workgroup of the bucket kernels

template <typename T> __device__ __forceinline__ T ldc(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(NT) void bucket_max_kernel(LoopState* __restrict__ st, const int32_t n, const int32_t* __restrict__ counts) {
  if (st->done) return;
  int best = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * NT + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * NT) {
    const int32_t c = counts[i];
    if (c >= 2) best = c > best ? c : best;  // rows that left the live set carry the degree DEAD
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int other = __shfl_xor(best, o);
    best            = other > best ? other : best;
  }
  if ((threadIdx.x & 63) == 0 && best > 0) atomicMax(&st->bucketMax, best);
}

// reversed row index ri = 0 is row n - 1: ascending ri = descending row
__device__ __forceinline__ bool in_bucket(const int64_t ri, const int32_t n, const int D, const int32_t* __restrict__ counts) {
  return ri < n && counts[n - 1 - ri] == D;
}

__global__ __launch_bounds__(NT) void bucket_count_kernel(const LoopState* __restrict__ st, const int32_t n,
                                                          const int32_t* __restrict__ counts, int32_t* __restrict__ blockCounts) {
  if (st->done) return;
  const int D = st->bucketMax;
  int       c = 0;
  if (D >= 2) {
    for (int k = 0; k < BUCKET_ROWS / NT; ++k) {
      c += in_bucket(static_cast<int64_t>(blockIdx.x) * BUCKET_ROWS + k * NT + threadIdx.x, n, D, counts) ? 1 : 0;
    }
  }
  __shared__ int total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  if (c) atomicAdd(&total, c);
  __syncthreads();
  if (threadIdx.x == 0) blockCounts[blockIdx.x] = total;
}

__global__ __launch_bounds__(NT) void bucket_scan_kernel(LoopState* __restrict__ st, int32_t* __restrict__ blockCounts, const int nBlocks) {
  if (st->done) return;
  // exclusive scan in place, one workgroup (nBlocks <= a few thousand)
  __shared__ int carry;
  __shared__ int wsum[NT / 64];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nBlocks; base += NT) {
    const int i = base + threadIdx.x;
    const int v = i < nBlocks ? blockCounts[i] : 0;
    int       x = v;  // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if ((threadIdx.x & 63) >= o) x += y;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < (threadIdx.x >> 6); ++w) before += wsum[w];
    if (i < nBlocks) blockCounts[i] = before + x - v;
    __syncthreads();
    if (threadIdx.x == NT - 1) carry = before + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    st->nCand     = carry;
    st->curDegree = st->bucketMax;
  }
}

__global__ __launch_bounds__(NT) void bucket_fill_kernel(const LoopState* __restrict__ st, const int32_t n,
                                                         const int32_t* __restrict__ counts, const int32_t* __restrict__ blockOffsets,
                                                         int32_t* __restrict__ cand) {
  if (st->done) return;
  const int D = st->bucketMax;
  if (D < 2) return;
  __shared__ int wbase[NT / 64];
  __shared__ int running;
  if (threadIdx.x == 0) running = blockOffsets[blockIdx.x];
  __syncthreads();
  for (int k = 0; k < BUCKET_ROWS / NT; ++k) {
    const int64_t  ri = static_cast<int64_t>(blockIdx.x) * BUCKET_ROWS + k * NT + threadIdx.x;
    const bool     f  = in_bucket(ri, n, D, counts);
    const uint64_t m  = __ballot(f);
    if ((threadIdx.x & 63) == 0) wbase[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    int before = running;
    for (int w = 0; w < (threadIdx.x >> 6); ++w) before += wbase[w];
    if (f) cand[before + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = static_cast<int32_t>(n - 1 - ri);
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < NT / 64; ++w) t += wbase[w];
      running += t;
    }
    __syncthreads();
  }
}

// Rounds off the current bucket, on ONE workgroup: harvest L, extract the centroid's cluster, subtract its members
// from their live neighbours and collect the rows that drop to degree 1; repeat until the bucket is exhausted.
// A row that leaves the live set gets the degree DEAD (hugely negative), so "alive" needs no array of its own:
// extraction is one atomicExch per neighbour (old > 0: a new member), subtraction one atomicSub (old == 2: the row is
// down to itself -> next round's harvest; dead rows just get a little more negative), the bucket test is degree == D.
// A round is then a chain of ~8 dependent L2 round trips: members and their CSR ranges are handed from the extraction
// to the subtraction through LDS.
constexpr int32_t DEAD        = -(1 << 30);
constexpr int     LNT          = 1024;  // threads of the round-loop workgroup: 64 members' lists are walked at a time
constexpr int     LOOP_MEMBERS = 2048;  // members staged in LDS; larger clusters take the list back from clusterIdx

__global__ __launch_bounds__(LNT) void sparse_loop_kernel(LoopState* __restrict__ st, const unsigned long long* __restrict__ offsets,
                                                         const int32_t* __restrict__ nbr, int32_t* __restrict__ counts,
                                                         int32_t* __restrict__ clusterIdx, int32_t* __restrict__ clusterOffsets,
                                                         int32_t* __restrict__ centroids, int32_t* __restrict__ L0,
                                                         int32_t* __restrict__ L1, int32_t* __restrict__ nL,
                                                         const int32_t* __restrict__ cand) {
  if (st->done) return;
  __shared__ int                sCount;   // members found / rows added to the next L
  __shared__ int                sMaxRow;
  __shared__ int                sPick;    // index into cand of this round's centroid
  __shared__ unsigned long long sBeg[LOOP_MEMBERS];
  __shared__ int                sLen[LOOP_MEMBERS];
  const int tid    = threadIdx.x;
  const int D      = st->curDegree;
  const int nCand  = st->nCand;
  int       parity = st->parity;
  int       nLcur  = nL[parity];
  int       front = st->front, back = st->back, nClusters = st->nClusters;
  if (tid == 0) {
    sCount  = 0;
    sMaxRow = -1;
  }
  __syncthreads();
  if (D < 2) {
    // no row of degree >= 2 is left: the highest row of L is this round's centroid (a cluster of one), the others
    // are harvested; nothing else can change afterwards
    const int32_t* Lcur = parity ? L1 : L0;
    for (int i = tid; i < nLcur; i += LNT) atomicMax(&sMaxRow, Lcur[i]);
    __syncthreads();
    const int last = sMaxRow;
    if (last >= 0) {
      for (int i = tid; i < nLcur; i += LNT) {
        const int r = Lcur[i];
        counts[r]   = DEAD;
        if (r != last) clusterIdx[back - atomicAdd(&sCount, 1)] = r;
      }
    }
    __syncthreads();
    if (tid == 0) {
      if (last >= 0) {
        clusterIdx[front]             = last;
        centroids[nClusters]          = last;
        clusterOffsets[nClusters + 1] = front + 1;
        st->front                     = front + 1;
        st->nClusters                 = nClusters + 1;
        st->back                      = back - sCount;
      }
      st->done = 1;
    }
    return;
  }
  int cursor = 0;
  for (;;) {
    // ---- centroid: first bucket entry at or after the cursor that still has degree D ----
    int centroid = -1;
    while (cursor < nCand) {
      if (tid == 0) sPick = 0x7fffffff;
      __syncthreads();
      const int idx = cursor + tid;
      if (idx < nCand && ldc(&counts[cand[idx]]) == D) atomicMin(&sPick, idx);
      __syncthreads();
      const int pick = sPick;
      __syncthreads();
      if (pick != 0x7fffffff) {
        centroid = cand[pick];
        cursor   = pick + 1;
        break;
      }
      cursor += LNT;
    }
    if (centroid < 0) break;  // bucket exhausted: the next epoch builds the next one

    int32_t* Lcur  = parity ? L1 : L0;
    int32_t* Lnext = parity ? L0 : L1;
    // harvest: rows that reached degree 1 in the previous round (they have no live neighbour, nobody touches them)
    for (int i = tid; i < nLcur; i += LNT) {
      const int r          = ldc(&Lcur[i]);
      counts[r]            = DEAD;
      clusterIdx[back - i] = r;
    }
    // members = live neighbours of the centroid; they and the centroid leave the live set
    const unsigned long long o0 = offsets[centroid], o1 = offsets[centroid + 1];
    if (tid == 0) {
      clusterIdx[front] = centroid;
      counts[centroid]  = DEAD;  // its own list does not contain it, nobody else reads it this round
      sBeg[0]           = o0;
      sLen[0]           = static_cast<int>(o1 - o0);
    }
    for (unsigned long long k = o0 + tid; k < o1; k += LNT) {
      const int j = nbr[k];
      if (atomicExch(&counts[j], DEAD) > 0) {  // was alive: a new member (a neighbour appears once in the list)
        const int slot             = 1 + atomicAdd(&sCount, 1);
        clusterIdx[front + slot]   = j;
        if (slot < LOOP_MEMBERS) {
          const unsigned long long a0 = offsets[j];
          sBeg[slot]                  = a0;
          sLen[slot]                  = static_cast<int>(offsets[j + 1] - a0);
        }
      }
    }
    __syncthreads();
    const int total = 1 + sCount;
    __syncthreads();
    if (tid == 0) sCount = 0;
    if (total > LOOP_MEMBERS) __threadfence();  // the overflow members are re-read from clusterIdx below
    __syncthreads();
    // subtract: every neighbour of a member loses one; 16 lanes per member
    const int g = tid >> 4, l = tid & 15;
    for (int q = g; q < total; q += LNT / 16) {
      unsigned long long a0;
      int                len;
      if (q < LOOP_MEMBERS) {
        a0  = sBeg[q];
        len = sLen[q];
      } else {
        const int m = ldc(&clusterIdx[front + q]);
        a0          = offsets[m];
        len         = static_cast<int>(offsets[m + 1] - a0);
      }
      for (int k = l; k < len; k += 16) {
        const int j = nbr[a0 + k];
        if (ldc(&counts[j]) > 0 && atomicSub(&counts[j], 1) == 2) Lnext[atomicAdd(&sCount, 1)] = j;  // only itself left: next round's harvest
      }
    }
    __threadfence();  // Lnext is read back (from L2) in the next round
    __syncthreads();
    if (tid == 0) {
      centroids[nClusters]          = centroid;
      clusterOffsets[nClusters + 1] = front + total;
    }
    front += total;
    back -= nLcur;
    ++nClusters;
    nLcur = sCount;
    parity ^= 1;
    __syncthreads();
    if (tid == 0) sCount = 0;
    __syncthreads();
  }
  if (tid == 0) {
    st->front      = front;
    st->back       = back;
    st->nClusters  = nClusters;
    st->lastMax    = D;
    st->parity     = parity;
    st->bucketMax  = 0;
    nL[parity]     = nLcur;
    nL[parity ^ 1] = 0;
  }
}

// ---- parallel rounds off a bucket ---------------------------------------------------------------------------------
// The one-workgroup loop above walks the bucket's candidates c_1 > c_2 > ... one round at a time (10 us per round,
// 20 183 rounds = 0.19 s of a 0.66 s call at N = 1M).  Which candidates become centroids can be decided for the whole
// bucket at once: processing c_j changes c_k's closed neighbourhood exactly when the two LIVE closed neighbourhoods
// intersect (c_k dies as a member, or one of its neighbours does and its degree drops below D), so the sequential
// loop selects the lexicographically first maximal independent set of the candidates under that conflict relation.
// Sweeps compute the same set: every undecided candidate stamps its index on its live closed neighbourhood with
// atomicMin (mark); a candidate that finds its own stamp on all of it has no undecided earlier rival — and no selected
// one either, that would already have changed its degree — and is selected (decide); the selected clusters' members die
// and their surviving neighbours lose one degree each (subtract, commutative); candidates whose degree is no longer D
// are rejected by the next mark.  A candidate can only be rejected by an EARLIER selected one (a later rival never sees
// its own stamp while the earlier one is undecided), so selected / rejected are exactly the sequential loop's decisions.
// The first undecided candidate is decided in every sweep: two sweeps run on the whole chip, whatever they leave
// undecided (dependency chains) is finished by one workgroup sweeping on.  All clusters of a bucket have exactly D
// members, so the clusters are emitted after the bucket is decided, at offsets that follow from the candidates' order:
// the output is identical to the sequential loop's, cluster order included.  Rows that drop to degree 1 need no harvest
// lists here: they are isolated from then on and join the singleton tail at the end; the sequential loop's last act —
// the highest row that reached degree 1 in the LAST round closes the greedy list as a cluster of one — is par_final_kernel.
constexpr int PG = 16;  // lanes per candidate
constexpr int32_t NO_OWNER = 0x7fffffff;

__device__ __forceinline__ bool group_all(const bool ok) {  // AND over the PG lanes of a candidate's group
  const uint64_t m     = __ballot(ok);
  const int      shift = (threadIdx.x & 63) & ~(PG - 1);
  return ((m >> shift) & ((1ull << PG) - 1ull)) == ((1ull << PG) - 1ull);
}

struct ParArgs {
  LoopState*                st;
  const unsigned long long* offsets;
  const int32_t*            nbr;
  int32_t*                  counts;
  const int32_t*            cand;
  int32_t*                  status;    // per candidate index: 0 undecided, 1 selected, -1 rejected
  int32_t*                  U[2];      // undecided lists (candidate indices)
  int32_t*                  owner;     // per row: smallest undecided candidate index whose live closed neighbourhood holds it
  int32_t*                  memberOf;  // per row: centroid of the cluster it joined (-1: none yet)
};

// `first`: the input list is the whole bucket 0..nCand-1; otherwise U[p].  Output list U[p ^ 1].
template <bool FENCED> __device__ __forceinline__ void par_mark(const ParArgs& a, const bool first, const int p, const int group,
                                                                 const int nGroups, const int l) {
  LoopState* st  = a.st;
  const int  D   = st->curDegree;
  const int  nIn = first ? st->nCand : (FENCED ? ldc(&st->nU[p]) : st->nU[p]);
  for (int i = group; i < nIn; i += nGroups) {
    const int k = first ? i : a.U[p][i];
    if (!first && a.status[k] != 0) continue;  // selected in the previous sweep
    const int c = a.cand[k];
    if (a.counts[c] != D) {  // dead, or lost a neighbour to an earlier cluster: it may come back in a later bucket
      if (l == 0) a.status[k] = -1;
      continue;
    }
    if (l == 0) {
      a.status[k] = 0;
      a.U[p ^ 1][atomicAdd(&st->nU[p ^ 1], 1)] = k;
      atomicMin(&a.owner[c], k);
    }
    const unsigned long long o0 = a.offsets[c], o1 = a.offsets[c + 1];
    for (unsigned long long e = o0 + l; e < o1; e += PG) {
      const int v = a.nbr[e];
      if (a.counts[v] > 0) atomicMin(&a.owner[v], k);
    }
  }
}

// list U[p] (what mark wrote); no degree is read here, so the members may die in the same step
template <bool FENCED> __device__ __forceinline__ void par_decide(const ParArgs& a, const int p, const int group, const int nGroups,
                                                                   const int l) {
  const int nIn = FENCED ? ldc(&a.st->nU[p]) : a.st->nU[p];
  for (int i0 = group - (group % (64 / PG)); i0 < nIn; i0 += nGroups) {  // the groups of a wave stay together (ballots)
    const int  i     = i0 + group % (64 / PG);
    const bool valid = i < nIn;
    const int  k     = valid ? a.U[p][i] : 0;
    const int  c     = valid ? a.cand[k] : 0;
    bool       ok    = valid;
    unsigned long long o0 = 0, o1 = 0;
    if (valid) {
      o0 = a.offsets[c];
      o1 = a.offsets[c + 1];
      if (l == 0 && a.owner[c] != k) ok = false;
      for (unsigned long long e = o0 + l; e < o1; e += PG) {
        const int o = a.owner[a.nbr[e]];
        if (o != NO_OWNER && o != k) ok = false;  // a live neighbour (it carries a stamp) that an earlier candidate claims
      }
    }
    const bool sel = group_all(ok || !valid) && valid;
    if (sel) {
      if (l == 0) {
        a.status[k] = 1;
        a.counts[c] = DEAD;
      }
      for (unsigned long long e = o0 + l; e < o1; e += PG) {
        const int v = a.nbr[e];
        if (a.owner[v] == k) {  // live when it was stamped: a member
          a.counts[v]   = DEAD;
          a.memberOf[v] = c;
        }
      }
    }
  }
}

// list U[p] again: take the stamps back; the clusters selected in this sweep subtract their members from the survivors
template <bool FENCED> __device__ __forceinline__ void par_subtract(const ParArgs& a, const int p, const int group, const int nGroups,
                                                                     const int l) {
  const int nIn = FENCED ? ldc(&a.st->nU[p]) : a.st->nU[p];
  for (int i = group; i < nIn; i += nGroups) {
    const int                k  = a.U[p][i];
    const int                c  = a.cand[k];
    const unsigned long long o0 = a.offsets[c], o1 = a.offsets[c + 1];
    if (l == 0) a.owner[c] = NO_OWNER;
    for (unsigned long long e = o0 + l; e < o1; e += PG) a.owner[a.nbr[e]] = NO_OWNER;
    if (a.status[k] != 1) continue;
    for (unsigned long long e = o0; e < o1; ++e) {
      const int v = a.nbr[e];
      if (a.memberOf[v] != c) continue;  // (uniform over the group)
      const unsigned long long m0 = a.offsets[v], m1 = a.offsets[v + 1];
      for (unsigned long long f = m0 + l; f < m1; f += PG) {
        const int j = a.nbr[f];
        if (a.counts[j] > 0) atomicSub(&a.counts[j], 1);  // every member is DEAD already: only survivors are touched
      }
    }
  }
}

__global__ __launch_bounds__(NT) void par_mark_kernel(const ParArgs a, const int first, const int p) {
  if (a.st->done || a.st->curDegree < 2) return;
  const int t = blockIdx.x * NT + threadIdx.x;
  par_mark<false>(a, first != 0, p, t / PG, static_cast<int>(gridDim.x) * NT / PG, t % PG);
}
__global__ __launch_bounds__(NT) void par_decide_kernel(const ParArgs a, const int p) {
  if (a.st->done || a.st->curDegree < 2) return;
  const int t = blockIdx.x * NT + threadIdx.x;
  par_decide<false>(a, p, t / PG, static_cast<int>(gridDim.x) * NT / PG, t % PG);
}
__global__ __launch_bounds__(NT) void par_subtract_kernel(const ParArgs a, const int p) {
  if (a.st->done || a.st->curDegree < 2) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) a.st->nU[p ^ 1] = 0;  // this sweep's input list: the next sweep's output
  const int t = blockIdx.x * NT + threadIdx.x;
  par_subtract<false>(a, p, t / PG, static_cast<int>(gridDim.x) * NT / PG, t % PG);
}
// what the grid sweeps left undecided: one workgroup sweeps on (its own writes are re-read after agent-scope fences)
__global__ __launch_bounds__(LNT) void par_finish_kernel(const ParArgs a, int p) {
  if (a.st->done || a.st->curDegree < 2) return;
  const int group = threadIdx.x / PG, nGroups = LNT / PG, l = threadIdx.x % PG;
  // every sweep decides at least the first undecided candidate, so nCand sweeps are an upper bound; the cap only keeps a bug
  // from spinning on the GPU box (an unfinished bucket then shows up as an accounting error on the host)
  const int maxSweeps = a.st->nCand + 2;
  for (int sweep = 0; sweep < maxSweeps; ++sweep) {
    if (ldc(&a.st->nU[p]) == 0) break;
    par_mark<true>(a, false, p, group, nGroups, l);
    __threadfence();
    __syncthreads();
    __threadfence();
    par_decide<true>(a, p ^ 1, group, nGroups, l);
    __threadfence();
    __syncthreads();
    __threadfence();
    if (threadIdx.x == 0) a.st->nU[p] = 0;
    par_subtract<true>(a, p ^ 1, group, nGroups, l);
    __threadfence();
    __syncthreads();
    __threadfence();
    p ^= 1;
  }
  __syncthreads();
  if (threadIdx.x == 0) a.st->nU[0] = a.st->nU[1] = 0;
}

// selected candidates of the bucket, in candidate order -> selList (the count / scan / fill pattern of the bucket kernels)
__global__ __launch_bounds__(NT) void sel_count_kernel(const LoopState* __restrict__ st, const int32_t* __restrict__ status,
                                                       int32_t* __restrict__ blockCounts) {
  if (st->done) return;
  const int nCand = st->curDegree >= 2 ? st->nCand : 0;
  int       c     = 0;
  for (int k = 0; k < BUCKET_ROWS / NT; ++k) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * BUCKET_ROWS + k * NT + threadIdx.x;
    c += (i < nCand && status[i] == 1) ? 1 : 0;
  }
  __shared__ int total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  if (c) atomicAdd(&total, c);
  __syncthreads();
  if (threadIdx.x == 0) blockCounts[blockIdx.x] = total;
}
__global__ __launch_bounds__(NT) void sel_scan_kernel(LoopState* __restrict__ st, int32_t* __restrict__ blockCounts, const int nBlocks) {
  if (st->done) return;
  __shared__ int carry;
  __shared__ int wsum[NT / 64];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nBlocks; base += NT) {
    const int i = base + threadIdx.x;
    const int v = i < nBlocks ? blockCounts[i] : 0;
    int       x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if ((threadIdx.x & 63) >= o) x += y;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < (threadIdx.x >> 6); ++w) before += wsum[w];
    if (i < nBlocks) blockCounts[i] = before + x - v;
    __syncthreads();
    if (threadIdx.x == NT - 1) carry = before + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int D = st->curDegree, nSel = carry;
    st->nSel        = nSel;
    st->emitFront   = st->front;
    st->emitCluster = st->nClusters;
    if (D >= 2) {
      st->front += nSel * D;  // every cluster of the bucket has exactly D members
      st->nClusters += nSel;
      st->lastMax = D;
    } else {
      st->done = 1;  // no row of degree >= 2 is left
    }
    st->bucketMax = 0;
  }
}
__global__ __launch_bounds__(NT) void sel_fill_kernel(const LoopState* __restrict__ st, const int32_t* __restrict__ status,
                                                      const int32_t* __restrict__ blockOffsets, int32_t* __restrict__ selList) {
  if (st->done || st->curDegree < 2) return;
  const int nCand = st->nCand;
  __shared__ int wbase[NT / 64];
  __shared__ int running;
  if (threadIdx.x == 0) running = blockOffsets[blockIdx.x];
  __syncthreads();
  for (int k = 0; k < BUCKET_ROWS / NT; ++k) {
    const int64_t  i = static_cast<int64_t>(blockIdx.x) * BUCKET_ROWS + k * NT + threadIdx.x;
    const bool     f = i < nCand && status[i] == 1;
    const uint64_t m = __ballot(f);
    if ((threadIdx.x & 63) == 0) wbase[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    int before = running;
    for (int w = 0; w < (threadIdx.x >> 6); ++w) before += wbase[w];
    if (f) selList[before + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = static_cast<int32_t>(i);
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < NT / 64; ++w) t += wbase[w];
      running += t;
    }
    __syncthreads();
  }
}
// cluster r of the bucket: centroid first, then its members (their order is canonicalised on the host)
__global__ __launch_bounds__(NT) void par_emit_kernel(const ParArgs a, const int32_t* __restrict__ selList, int32_t* __restrict__ clusterIdx,
                                                      int32_t* __restrict__ clusterOffsets, int32_t* __restrict__ centroids) {
  const LoopState* st = a.st;
  if (st->done || st->curDegree < 2) return;
  const int D = st->curDegree, nSel = st->nSel, front = st->emitFront, first = st->emitCluster;
  const int t = blockIdx.x * NT + threadIdx.x, l = t % PG, nGroups = static_cast<int>(gridDim.x) * NT / PG;
  const int shift = (threadIdx.x & 63) & ~(PG - 1);
  for (int r0 = t / PG - (t / PG) % (64 / PG); r0 < nSel; r0 += nGroups) {  // the groups of a wave stay together (ballots)
    const int  r     = r0 + (t / PG) % (64 / PG);
    const bool valid = r < nSel;
    const int  c     = valid ? a.cand[selList[r]] : 0;
    const int  base  = front + r * D;
    if (valid && l == 0) {
      clusterIdx[base]              = c;
      centroids[first + r]          = c;
      clusterOffsets[first + r + 1] = base + D;
    }
    const unsigned long long o0 = valid ? a.offsets[c] : 0, o1 = valid ? a.offsets[c + 1] : 0;
    const unsigned long long steps = (o1 - o0 + PG - 1) / PG;
    unsigned long long maxSteps = steps;  // all groups of the wave take the same number of ballot steps
#pragma unroll
    for (int o = PG; o < 64; o <<= 1) {
      const unsigned long long other = __shfl_xor(maxSteps, o);
      maxSteps                       = other > maxSteps ? other : maxSteps;
    }
    int written = 1;
    for (unsigned long long q = 0; q < maxSteps; ++q) {
      const unsigned long long e = o0 + q * PG + l;
      const int                v = e < o1 ? a.nbr[e] : -1;
      const bool               f = v >= 0 && a.memberOf[v] == c;
      const uint64_t           m = (__ballot(f) >> shift) & ((1ull << PG) - 1ull);
      if (f) clusterIdx[base + written + __popcll(m & ((1ull << l) - 1ull))] = v;
      written += __popcll(m);
    }
  }
}
// The sequential loop's last act: when no row of degree >= 2 is left, the highest of the rows that reached degree 1 in the
// LAST round (= the degree-1 rows next to a member of the last cluster; with no cluster at all: every degree-1 row) closes
// the greedy list as a cluster of one.  Every other degree-1 row was, or would be, harvested into the singleton tail.
__global__ __launch_bounds__(LNT) void par_final_kernel(const ParArgs a, const int32_t n, int32_t* __restrict__ clusterIdx,
                                                        int32_t* __restrict__ clusterOffsets, int32_t* __restrict__ centroids) {
  LoopState* st = a.st;
  __shared__ int sMaxRow;
  if (threadIdx.x == 0) sMaxRow = -1;
  __syncthreads();
  const int nClusters = st->nClusters, front = st->front;
  if (nClusters > 0) {
    const int lo = clusterOffsets[nClusters - 1], hi = clusterOffsets[nClusters];
    const int g = threadIdx.x / PG, l = threadIdx.x % PG;
    for (int q = lo + g; q < hi; q += LNT / PG) {
      const int                m  = clusterIdx[q];
      const unsigned long long m0 = a.offsets[m], m1 = a.offsets[m + 1];
      for (unsigned long long f = m0 + l; f < m1; f += PG) {
        const int j = a.nbr[f];
        if (a.counts[j] == 1) atomicMax(&sMaxRow, j);
      }
    }
  } else {
    for (int i = threadIdx.x; i < n; i += LNT)
      if (a.counts[i] == 1) atomicMax(&sMaxRow, i);
  }
  __syncthreads();
  if (threadIdx.x == 0 && sMaxRow >= 0) {
    const int last                = sMaxRow;
    clusterIdx[front]             = last;
    centroids[nClusters]          = last;
    clusterOffsets[nClusters + 1] = front + 1;
    a.counts[last]                = DEAD;
    st->front                     = front + 1;
    st->nClusters                 = nClusters + 1;
  }
}
__global__ void fill_i32_kernel(int32_t* __restrict__ p, const int64_t n, const int32_t v) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// rows that never had a neighbour (degree 0: all-zero fingerprints) join the singleton tail; after the parallel rounds
// (withOnes) so do the rows left at degree 1, which the sequential loop harvests round by round
__global__ void sparse_leftover_kernel(LoopState* __restrict__ st, const int32_t n, const int32_t* __restrict__ counts,
                                       int32_t* __restrict__ clusterIdx, const int withOnes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (counts[i] == 0 || (withOnes && counts[i] == 1))) clusterIdx[atomicSub(&st->back, 1)] = i;
}

// ---- popcount-sorted first pass (NVMK_BUTINA_SORT=0 turns it off) ---------------------------------------
// Tanimoto(a, b) <= min(|a|, |b|) / max(|a|, |b|): with the rows sorted by popcount, whole tiles whose popcount bands
// are further apart than the threshold hold no neighbour pair and the count kernel skips them (fp4::CountArgs::bandSkip).
// Only the all-pairs pass runs on the sorted copy; pairs and degrees are mapped back to the original row numbers, so
// everything downstream (ties toward the highest ORIGINAL row, output order) is unchanged.
__global__ void row_popcount_kernel(const uint32_t* __restrict__ x, const int64_t n, const int W, int32_t* __restrict__ pop) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* r = reinterpret_cast<const uint4*>(x + i * W);
  int          c = 0;
  for (int w = 0; w < W / 4; ++w) {
    const uint4 v = r[w];
    c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
  }
  pop[i] = c;
}
__global__ void remap_edges_kernel(int2* __restrict__ edges, const unsigned long long nEdges, const int32_t* __restrict__ perm) {
  for (unsigned long long e = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < nEdges;
       e += static_cast<unsigned long long>(gridDim.x) * blockDim.x) {
    const int2 p = edges[e];
    edges[e]     = make_int2(perm[p.x], perm[p.y]);
  }
}
__global__ void scatter_counts_kernel(const int32_t* __restrict__ sorted, const int32_t* __restrict__ perm, const int64_t n,
                                      int32_t* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[perm[i]] = sorted[i];
}
inline bool sorted_first_pass() {  // default on; NVMK_BUTINA_SORT=0 keeps the input order
  return opt::get(opt::kButinaSort).s[0] != '0';
}

// Buffers of a fused-Butina run: one scratch block | state | nAliveNext[2] | counts | alive[2] | removed | clusterIndices |
// offsets | centroids.
struct RoundBuffers {
  LoopState* st;
  int32_t *  nAliveNext, *counts, *alive0, *alive1, *removed, *clusterIdx, *offsets, *centroids;
};

inline int alloc_round_buffers(StreamScratch& mem, const int64_t N, hipStream_t stream, RoundBuffers& b) {
  const size_t n    = static_cast<size_t>(N);
  const size_t ints = 64 + n * 7 + 8;
  NVMK_HIP_CHECK(mem.alloc(ints * sizeof(int32_t), stream));
  auto* base   = mem.as<int32_t>();
  b.st         = reinterpret_cast<LoopState*>(base);
  b.nAliveNext = base + 32;  // [2]
  b.counts     = base + 64;
  b.alive0     = b.counts + n;
  b.alive1     = b.alive0 + n;
  b.removed    = b.alive1 + n;
  b.clusterIdx = b.removed + n;
  b.offsets    = b.clusterIdx + n;  // n + 1 entries
  b.centroids  = b.offsets + n + 1;
  NVMK_HIP_CHECK(hipMemsetAsync(base, 0, (64 + n) * sizeof(int32_t), stream));  // state + nAliveNext + counts
  NVMK_HIP_CHECK(hipMemsetAsync(b.offsets, 0, sizeof(int32_t), stream));
  hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(1), 0, stream, b.st, static_cast<int32_t>(N));
  hipLaunchKernelGGL(iota_kernel, dim3(static_cast<unsigned>(ceil_div<int64_t>(N, 256))), dim3(256), 0, stream, b.alive0, N);
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

// The round loop on the sparse neighbour graph: `edges` holds every neighbour pair once (original row numbers), b.counts
// the degrees.  CSR (degrees -> exclusive scan -> fill), then epochs of the device-side loop.  On return `snap` is the
// final loop state with nAlive = 0 (leftovers already sit in the singleton tail).
inline int graph_rounds(const RoundBuffers& b, const int64_t N, const int2* edges, const unsigned long long nEdges, hipStream_t stream,
                        LoopState& snap) {
  const size_t n = static_cast<size_t>(N);
  LoopState*   st = b.st;
  int32_t *    counts = b.counts, *clusterIdx = b.clusterIdx, *offsets = b.offsets, *centroids = b.centroids, *nAliveNext = b.nAliveNext;
  // CSR: degrees -> exclusive scan -> fill
  StreamScratch csrMem, scanTmp;
  const size_t  offBytes = (n + 1) * sizeof(unsigned long long);
  const size_t  nbrBytes = std::max<size_t>(1, static_cast<size_t>(2 * nEdges)) * sizeof(int32_t);
  // layout: deg[n+1] | offsets[n+1] | cursor[n] (u32) | L0[n] | L1[n] | nbr[2E]
  const size_t bytes = 2 * offBytes + n * 4 * 3 + nbrBytes + 64;
  NVMK_HIP_CHECK(csrMem.alloc(bytes, stream));
  auto* deg     = csrMem.as<unsigned long long>();
  auto* offs64  = deg + (n + 1);
  auto* cursor  = reinterpret_cast<unsigned int*>(offs64 + (n + 1));
  auto* L0      = reinterpret_cast<int32_t*>(cursor + n);
  auto* L1      = L0 + n;
  auto* nbr     = L1 + n;
  NVMK_HIP_CHECK(hipMemsetAsync(deg, 0, 2 * offBytes + n * 4, stream));  // deg, offsets, cursor
  const unsigned eBlocks = static_cast<unsigned>(std::min<unsigned long long>(ceil_div<unsigned long long>(std::max<unsigned long long>(nEdges, 1), 256), 65535));
  if (nEdges > 0) {
    hipLaunchKernelGGL(edge_degree_kernel, dim3(eBlocks), dim3(256), 0, stream, edges, nEdges, deg);
    NVMK_LAUNCH_CHECK();
  }
  size_t tmpBytes = 0;
  NVMK_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, deg, offs64, static_cast<int>(n + 1), stream));
  NVMK_HIP_CHECK(scanTmp.alloc(tmpBytes, stream));
  NVMK_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(scanTmp.ptr, tmpBytes, deg, offs64, static_cast<int>(n + 1), stream));
  if (nEdges > 0) {
    hipLaunchKernelGGL(csr_fill_kernel, dim3(eBlocks), dim3(256), 0, stream, edges, nEdges, offs64, cursor, nbr);
    NVMK_LAUNCH_CHECK();
  }
  int32_t* nL = nAliveNext;  // [2], zeroed above with the state block
  hipLaunchKernelGGL(sparse_init_kernel, dim3(static_cast<unsigned>(ceil_div<int64_t>(N, 256))), dim3(256), 0, stream,
                     static_cast<int32_t>(N), counts, L0, &nL[0]);
  NVMK_LAUNCH_CHECK();
  // epochs: build the bucket of the current maximal degree (4 small kernels), then run its rounds — as parallel sweeps
  // over the whole bucket (default) or one after the other on a persistent workgroup (NVMK_BUTINA_ROUNDS=serial)
  const bool    serial   = serial_rounds();
  StreamScratch bucketMem;
  const int     nBuckets = static_cast<int>(ceil_div<int64_t>(N, BUCKET_ROWS));
  NVMK_HIP_CHECK(bucketMem.alloc(((serial ? 1 : 7) * n + nBuckets + 16) * sizeof(int32_t), stream));
  int32_t*       cand        = bucketMem.as<int32_t>();
  int32_t*       blockCounts = cand + (serial ? 1 : 7) * n;
  const unsigned maxBlocks   = static_cast<unsigned>(std::min<int64_t>(ceil_div<int64_t>(N, NT * 4), 1024));
  ParArgs        pa{};
  int32_t*       selList = nullptr;
  if (!serial) {
    pa.st       = st;
    pa.offsets  = offs64;
    pa.nbr      = nbr;
    pa.counts   = counts;
    pa.cand     = cand;
    pa.status   = cand + n;
    pa.U[0]     = cand + 2 * n;
    pa.U[1]     = cand + 3 * n;
    pa.owner    = cand + 4 * n;
    pa.memberOf = cand + 5 * n;
    selList     = cand + 6 * n;
    const unsigned fb = static_cast<unsigned>(ceil_div<int64_t>(2 * N, 256));
    hipLaunchKernelGGL(fill_i32_kernel, dim3(fb), dim3(256), 0, stream, pa.owner, static_cast<int64_t>(N), NO_OWNER);
    hipLaunchKernelGGL(fill_i32_kernel, dim3(fb), dim3(256), 0, stream, pa.memberOf, static_cast<int64_t>(N), -1);
    NVMK_LAUNCH_CHECK();
  }
  const unsigned sweepBlocks = static_cast<unsigned>(std::min<int64_t>(std::max<int64_t>(ceil_div<int64_t>(N * PG, NT), 1), 2048));
  for (int64_t guard = 0;; ++guard) {
    // every epoch with a non-empty bucket forms at least one cluster and the empty one ends the loop, so N epochs
    // are an upper bound; anything beyond is a bug and must not spin on the GPU box
    NVMK_REQUIRE(guard * 4 <= N + 8, "fused butina: the round loop did not terminate (internal error)");
    for (int e = 0; e < 4; ++e) {
      hipLaunchKernelGGL(bucket_max_kernel, dim3(maxBlocks), dim3(NT), 0, stream, st, static_cast<int32_t>(N), counts);
      hipLaunchKernelGGL(bucket_count_kernel, dim3(nBuckets), dim3(NT), 0, stream, st, static_cast<int32_t>(N), counts, blockCounts);
      hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(NT), 0, stream, st, blockCounts, nBuckets);
      hipLaunchKernelGGL(bucket_fill_kernel, dim3(nBuckets), dim3(NT), 0, stream, st, static_cast<int32_t>(N), counts, blockCounts, cand);
      if (serial) {
        hipLaunchKernelGGL(sparse_loop_kernel, dim3(1), dim3(LNT), 0, stream, st, offs64, nbr, counts, clusterIdx, offsets, centroids,
                           L0, L1, nL, cand);
        continue;
      }
      // sweep 0 reads the whole bucket and writes U[1]; sweep 1 reads U[1] and writes U[0]; the finishing workgroup goes on from U[0]
      hipLaunchKernelGGL(par_mark_kernel, dim3(sweepBlocks), dim3(NT), 0, stream, pa, 1, 0);
      hipLaunchKernelGGL(par_decide_kernel, dim3(sweepBlocks), dim3(NT), 0, stream, pa, 1);
      hipLaunchKernelGGL(par_subtract_kernel, dim3(sweepBlocks), dim3(NT), 0, stream, pa, 1);
      hipLaunchKernelGGL(par_mark_kernel, dim3(sweepBlocks), dim3(NT), 0, stream, pa, 0, 1);
      hipLaunchKernelGGL(par_decide_kernel, dim3(sweepBlocks), dim3(NT), 0, stream, pa, 0);
      hipLaunchKernelGGL(par_subtract_kernel, dim3(sweepBlocks), dim3(NT), 0, stream, pa, 0);
      hipLaunchKernelGGL(par_finish_kernel, dim3(1), dim3(LNT), 0, stream, pa, 0);
      hipLaunchKernelGGL(sel_count_kernel, dim3(nBuckets), dim3(NT), 0, stream, st, pa.status, blockCounts);
      hipLaunchKernelGGL(sel_scan_kernel, dim3(1), dim3(NT), 0, stream, st, blockCounts, nBuckets);
      hipLaunchKernelGGL(sel_fill_kernel, dim3(nBuckets), dim3(NT), 0, stream, st, pa.status, blockCounts, selList);
      hipLaunchKernelGGL(par_emit_kernel, dim3(sweepBlocks), dim3(NT), 0, stream, pa, selList, clusterIdx, offsets, centroids);
    }
    NVMK_LAUNCH_CHECK();
    NVMK_HIP_CHECK(hipMemcpyAsync(&snap, st, sizeof(snap), hipMemcpyDeviceToHost, stream));
    NVMK_HIP_CHECK(hipStreamSynchronize(stream));
    if (snap.done) break;
  }
  if (!serial) {
    hipLaunchKernelGGL(par_final_kernel, dim3(1), dim3(LNT), 0, stream, pa, static_cast<int32_t>(N), clusterIdx, offsets, centroids);
  }
  hipLaunchKernelGGL(sparse_leftover_kernel, dim3(static_cast<unsigned>(ceil_div<int64_t>(N, 256))), dim3(256), 0, stream,
                     st, static_cast<int32_t>(N), counts, clusterIdx, serial ? 0 : 1);
  NVMK_LAUNCH_CHECK();
  NVMK_HIP_CHECK(hipMemcpyAsync(&snap, st, sizeof(snap), hipMemcpyDeviceToHost, stream));
  NVMK_HIP_CHECK(hipStreamSynchronize(stream));  // csrMem / scanTmp are released in stream order after this
  snap.nAlive = 0;                               // leftovers already sit in the singleton tail
  return NVMK_OK;
}

// Read the clusters back and canonicalise on the host (member order from atomics is unspecified).
inline int read_back(const RoundBuffers& b, const int64_t N, const LoopState& snap, int32_t* h_idx, int64_t* h_offsets,
                     int32_t* h_centroids, int64_t* n_clusters, hipStream_t stream) {
  const size_t   n = static_cast<size_t>(N);
  const int32_t *clusterIdx = b.clusterIdx, *offsets = b.offsets, *centroids = b.centroids, *alive0 = b.alive0, *alive1 = b.alive1;
  // ---- read back and canonicalise on the host (member order from atomics is unspecified) ----
  const int64_t nGreedy = snap.nClusters;
  // (the loop state comes from the device: nothing below may index with it unchecked — the caller's arrays hold N entries)
  if (nGreedy < 0 || nGreedy > N || snap.nAlive < 0 || snap.nAlive > N || snap.back < -1 || snap.back >= N) {
    set_last_error("fused butina: internal accounting error (loop state: %lld clusters, %lld alive, tail at %lld of %lld rows)",
                   (long long)nGreedy, (long long)snap.nAlive, (long long)snap.back, (long long)N);
    return NVMK_ERR_INTERNAL;
  }
  std::vector<int32_t> idx(n), offs(static_cast<size_t>(nGreedy) + 1), cent(static_cast<size_t>(nGreedy));
  NVMK_HIP_CHECK(hipMemcpyAsync(idx.data(), clusterIdx, n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  NVMK_HIP_CHECK(hipMemcpyAsync(offs.data(), offsets, offs.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  if (nGreedy > 0) {
    NVMK_HIP_CHECK(hipMemcpyAsync(cent.data(), centroids, cent.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  }
  // rows still alive when the loop stopped have degree 0 (all-zero fingerprints)
  const int32_t* aliveFinal = snap.finalParity ? alive1 : alive0;
  std::vector<int32_t> leftovers(static_cast<size_t>(snap.nAlive));
  if (snap.nAlive > 0) {
    NVMK_HIP_CHECK(hipMemcpyAsync(leftovers.data(), aliveFinal, leftovers.size() * sizeof(int32_t), hipMemcpyDeviceToHost,
                                  stream));
  }
  NVMK_HIP_CHECK(hipStreamSynchronize(stream));

  // A cluster's entries on the device: the centroid once, somewhere, and its members in the order the atomics fell; on the
  // host: centroid first, members ascending.  Sizes carry over, so every cluster's place is known up front and the clusters
  // are copied and sorted by a few host threads (one thread took 35 ms for the 20 183 clusters of the 1M-row benchmark, 7 % of
  // the call).
  {
    // every row is in exactly one place: the greedy clusters' entries, the singleton tail, the leftovers
    bool ok = offs[0] >= 0 && offs[static_cast<size_t>(nGreedy)] <= N;
    for (int64_t k = 0; ok && k < nGreedy; ++k) ok = offs[static_cast<size_t>(k) + 1] > offs[static_cast<size_t>(k)];
    const int64_t accounted = static_cast<int64_t>(offs[static_cast<size_t>(nGreedy)]) - offs[0] + (N - (snap.back + 1)) + snap.nAlive;
    if (!ok || accounted != N || nGreedy + (N - (snap.back + 1)) + snap.nAlive > N) {
      set_last_error("fused butina: internal accounting error (%lld of %lld rows assigned)", (long long)accounted, (long long)N);
      return NVMK_ERR_INTERNAL;
    }
  }
  h_offsets[0] = 0;
  for (int64_t k = 0; k < nGreedy; ++k) {
    h_offsets[k + 1] = h_offsets[k] + (offs[static_cast<size_t>(k) + 1] - offs[static_cast<size_t>(k)]);
    h_centroids[k]   = cent[static_cast<size_t>(k)];
  }
  int64_t pos = nGreedy > 0 ? h_offsets[nGreedy] : 0;
  {
    std::atomic<int64_t> next{0};
    std::atomic<int>     bad{0};
    constexpr int64_t    kBlock = 128;  // clusters per work item
    auto                 worker = [&] {
      for (;;) {
        const int64_t k0 = next.fetch_add(kBlock);
        if (k0 >= nGreedy) return;
        for (int64_t k = k0; k < std::min(nGreedy, k0 + kBlock); ++k) {
          const int32_t c   = cent[static_cast<size_t>(k)];
          int32_t*      dst = h_idx + h_offsets[k];
          int64_t       m   = 0;
          dst[m++]          = c;
          const int64_t size = h_offsets[k + 1] - h_offsets[k];
          for (int32_t q = offs[static_cast<size_t>(k)]; q < offs[static_cast<size_t>(k) + 1]; ++q) {
            if (idx[static_cast<size_t>(q)] != c && m < size) dst[m++] = idx[static_cast<size_t>(q)];
          }
          if (m != size) bad.store(1);  // the centroid missing from its cluster, or there twice
          std::sort(dst + 1, dst + m);
        }
      }
    };
    const int64_t items   = (nGreedy + kBlock - 1) / kBlock;
    const int     threads = static_cast<int>(std::min<int64_t>(items, std::min(16u, std::max(1u, std::thread::hardware_concurrency()))));
    if (threads <= 1) {
      worker();
    } else {
      // thread creation can throw (thread limits, a cgroup's pid cap): nothing may unwind through the C ABI, so the threads
      // that did start are joined and the calling thread takes whatever work is left
      std::vector<std::thread> pool;
      try {
        pool.reserve(static_cast<size_t>(threads));
        for (int t = 0; t + 1 < threads; ++t) pool.emplace_back(worker);
      } catch (...) {
      }
      worker();
      for (std::thread& t : pool) t.join();
    }
    if (bad.load() != 0) {
      set_last_error("fused butina: internal accounting error (a cluster without exactly one copy of its centroid)");
      return NVMK_ERR_INTERNAL;
    }
  }
  std::vector<int32_t> singles(idx.begin() + (snap.back + 1), idx.end());
  singles.insert(singles.end(), leftovers.begin(), leftovers.end());
  std::sort(singles.begin(), singles.end());
  int64_t k = nGreedy;
  for (const int32_t s : singles) {
    h_idx[pos++]   = s;
    h_centroids[k] = s;
    h_offsets[++k] = pos;
  }
  if (pos != N) {
    set_last_error("fused butina: internal accounting error (%lld of %lld rows assigned)", (long long)pos, (long long)N);
    return NVMK_ERR_INTERNAL;
  }
  *n_clusters = k;
  return NVMK_OK;
}

template <int METRIC>
int fused_impl(const uint32_t* d_x, int64_t N, int fpBits, double cutoff, int32_t* h_idx, int64_t* h_offsets,
               int32_t* h_centroids, int64_t* n_clusters, hipStream_t stream) {
  const int   W   = fpBits / 32;
  const float thr = static_cast<float>(1.0 - cutoff);  // clustering.py:149
  NVMK_REQUIRE(N <= 0x7fffffffLL, "fused butina: N too large (%lld)", (long long)N);
  NVMK_REQUIRE(W % 4 == 0 && reinterpret_cast<uintptr_t>(d_x) % 16 == 0,
               "fused butina: fp_bits must be a multiple of 128 and the matrix 16-byte aligned");

  StreamScratch tableMem, mem;
  CountPlan     plan;
  int           rc = make_plan(plan, tableMem, METRIC, fpBits, thr, stream);
  if (rc != NVMK_OK) return rc;

  const size_t n = static_cast<size_t>(N);
  RoundBuffers rb{};
  rc = alloc_round_buffers(mem, N, stream, rb);
  if (rc != NVMK_OK) return rc;
  LoopState* st         = rb.st;
  int32_t *  nAliveNext = rb.nAliveNext, *counts = rb.counts, *alive0 = rb.alive0, *alive1 = rb.alive1, *removed = rb.removed,
          *clusterIdx = rb.clusterIdx, *offsets = rb.offsets, *centroids = rb.centroids;

  // Matrix-core path: expand the whole set to FP4 once (4x the packed bytes); every counting pass then
  // gathers rows of it through the alive / removed index lists.  NVMK_SIM_PATH=valu keeps the v_bcnt kernels.
  const bool    useMfma = !force_valu() && (force_mfma() || N >= 2048);
  StreamScratch fp4Mem;
  fp4::Prepared PX{};
  // popcount-sorted first pass: only with the sparse-graph loop (it consumes the emitted pairs) and Tanimoto
  const bool    sortPass = useMfma && sorted_first_pass() && !dense_rounds() && METRIC == NVMK_METRIC_TANIMOTO && thr > 0.0f;
  StreamScratch sortMem, sortTmp;
  int32_t*      perm = nullptr;  // sorted position -> original row
  if (useMfma) {
    NVMK_HIP_CHECK(fp4Mem.alloc(fp4::layout(N, fpBits).bytes, stream));
    if (sortPass) {
      NVMK_HIP_CHECK(sortMem.alloc(3 * n * sizeof(int32_t), stream));
      int32_t* pop = sortMem.as<int32_t>();
      int32_t* popSorted = pop + n;
      perm               = popSorted + n;
      hipLaunchKernelGGL(row_popcount_kernel, dim3(static_cast<unsigned>(ceil_div<int64_t>(N, 256))), dim3(256), 0, stream, d_x, N, W, pop);
      NVMK_LAUNCH_CHECK();
      size_t tmpBytes = 0;
      NVMK_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmpBytes, pop, popSorted, alive0, perm, static_cast<int>(N), 0, 32, stream));
      NVMK_HIP_CHECK(sortTmp.alloc(tmpBytes, stream));
      NVMK_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(sortTmp.ptr, tmpBytes, pop, popSorted, alive0, perm, static_cast<int>(N), 0, 32, stream));
    }
    rc = fp4::prepare(d_x, perm, N, fpBits, fp4Mem.ptr, stream);
    if (rc != NVMK_OK) return rc;
    PX = fp4::view(fp4Mem.ptr, N, fpBits);
  }
  auto count_pass = [&](const int32_t* xRows, int64_t nX, const int32_t* nXdev, const int32_t* yRows, int64_t nY,
                        const int32_t* nYdev, int sign, bool symmetric) -> int {
    if (useMfma) {
      fp4::CountArgs a{};
      a.metric    = METRIC;
      a.thr       = thr;
      a.table     = plan.table;
      a.tableF    = plan.tableF;
      a.sign      = sign;
      a.xRows     = xRows;
      a.nX        = nX;
      a.nXdev     = nXdev;
      a.yRows     = yRows;
      a.nY        = nY;
      a.nYdev     = nYdev;
      a.symmetric = symmetric;
      return fp4::launch_counts(a, PX, PX, counts, stream);
    }
    return launch_counts(plan, d_x, xRows, nX, nXdev, d_x, yRows, nY, nYdev, sign, counts, stream);
  };

  // first pass: all-vs-all degrees (upper triangle of tiles only on the matrix-core path); the matrix-core kernel
  // also emits the neighbour pairs for the sparse-graph round loop
  const unsigned long long edgeCap =
    useMfma && !dense_rounds() ? std::min<unsigned long long>(static_cast<unsigned long long>(N) * 128ull, (1ull << 30) - 1ull) : 0ull;
  StreamScratch edgeMem;
  int2*               edges      = nullptr;
  unsigned long long* edgeCursor = nullptr;
  if (edgeCap > 0) {
    NVMK_HIP_CHECK(edgeMem.alloc(256 + edgeCap * sizeof(int2), stream));
    edgeCursor = edgeMem.as<unsigned long long>();
    edges      = reinterpret_cast<int2*>(edgeMem.as<char>() + 256);
    NVMK_HIP_CHECK(hipMemsetAsync(edgeCursor, 0, 256, stream));
    fp4::CountArgs a{};
    a.metric       = METRIC;
    a.thr          = thr;
    a.table        = plan.table;
    a.tableF       = plan.tableF;
    a.sign         = +1;
    a.nX           = N;
    a.nY           = N;
    a.symmetric    = true;
    a.edges        = edges;
    a.edgeCursor   = edgeCursor;
    a.edgeCapacity = edgeCap;
    a.bandSkip     = sortPass;
    rc             = fp4::launch_counts(a, PX, PX, counts, stream);
  } else {
    rc = count_pass(nullptr, N, nullptr, nullptr, N, nullptr, +1, true);
  }
  if (rc != NVMK_OK) return rc;
  if (sortPass && edgeCap > 0) {
    // back to original row numbers: degrees now, pairs once their number is known; the FP4 copy goes back to the
    // original order too (only the dense-round fallback reads it again)
    hipLaunchKernelGGL(scatter_counts_kernel, dim3(static_cast<unsigned>(ceil_div<int64_t>(N, 256))), dim3(256), 0, stream, counts, perm, N,
                       removed);
    NVMK_LAUNCH_CHECK();
    NVMK_HIP_CHECK(hipMemcpyAsync(counts, removed, n * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
  }

  LoopState snap{};
  bool      sparseDone = false;
  if (edgeCap > 0) {
    unsigned long long nEdges = 0;
    NVMK_HIP_CHECK(hipMemcpyAsync(&nEdges, edgeCursor, sizeof(nEdges), hipMemcpyDeviceToHost, stream));
    NVMK_HIP_CHECK(hipStreamSynchronize(stream));
    if (sortPass && nEdges > edgeCap) {  // dense-round fallback: it gathers rows of the FP4 copy by ORIGINAL row number
      rc = fp4::prepare(d_x, nullptr, N, fpBits, fp4Mem.ptr, stream);
      if (rc != NVMK_OK) return rc;
    }
    if (nEdges <= edgeCap) {  // otherwise: too dense for the edge buffer, the dense rounds below handle it
      if (sortPass && nEdges > 0) {
        const unsigned eb = static_cast<unsigned>(std::min<unsigned long long>(ceil_div<unsigned long long>(nEdges, 256), 65535));
        hipLaunchKernelGGL(remap_edges_kernel, dim3(eb), dim3(256), 0, stream, edges, nEdges, perm);
        NVMK_LAUNCH_CHECK();
      }
      rc = graph_rounds(rb, N, edges, nEdges, stream, snap);
      if (rc != NVMK_OK) return rc;
      sparseDone  = true;
    }
  }

  // dense round loop (small N, VALU path, or a graph too dense for the edge buffer), enqueued in batches; the host only
  // reads the state word between batches
  const auto* x4        = reinterpret_cast<const uint4*>(d_x);
  int64_t     aliveHost = N;  // upper bound on the device-side nAlive
  int64_t     maxDegree = N;  // upper bound on any later cluster size (degrees only decrease)
  int64_t     round     = 0;
  int         batch = 1;      // first sync after one round: learns the real max degree
  for (; !sparseDone;) {
    for (int b = 0; b < batch; ++b, ++round) {
      const int      parity   = static_cast<int>(round & 1);
      const int32_t* aliveIn  = parity ? alive1 : alive0;
      int32_t*       aliveOut = parity ? alive0 : alive1;
      const unsigned rowBlocks =
        static_cast<unsigned>(std::min<int64_t>(ceil_div<int64_t>(aliveHost, NT), 4096));
      const unsigned exBlocks = static_cast<unsigned>(ceil_div<int64_t>(aliveHost, EXTRACT_ROWS));
      hipLaunchKernelGGL(argmax_kernel, dim3(rowBlocks), dim3(NT), 0, stream, st, aliveIn, counts, parity);
      hipLaunchKernelGGL((extract_kernel<METRIC>), dim3(exBlocks), dim3(NT), 0, stream, st, x4, W / 4, aliveIn,
                         aliveOut, &nAliveNext[parity], removed, counts, clusterIdx, thr, parity);
      hipLaunchKernelGGL(finish_round_kernel, dim3(1), dim3(1), 0, stream, st, &nAliveNext[parity], offsets, centroids,
                         parity);
      // subtract the removed members' contribution from the survivors (device-side sizes)
      rc = count_pass(aliveOut, aliveHost, &st->nAlive, removed, maxDegree, &st->nRemoved, -1, false);
      if (rc != NVMK_OK) return rc;
      hipLaunchKernelGGL(reset_round_kernel, dim3(1), dim3(1), 0, stream, st, &nAliveNext[parity ^ 1]);
      NVMK_LAUNCH_CHECK();
    }
    NVMK_HIP_CHECK(hipMemcpyAsync(&snap, st, sizeof(snap), hipMemcpyDeviceToHost, stream));
    NVMK_HIP_CHECK(hipStreamSynchronize(stream));
    if (snap.done) break;
    aliveHost = snap.nAlive;
    maxDegree = std::max<int64_t>(1, snap.lastMax);
    batch     = 32;
  }

  return read_back(rb, N, snap, h_idx, h_offsets, h_centroids, n_clusters, stream);
}

// ---- row-sharded first pass ---------------------------------------------------------------
// Shard `shard` of `nShards` evaluates its band of tile rows of the symmetric all-pairs pass (bands of equal AREA of the
// upper triangle, so the shards do equal work): partial degrees of ALL rows (row and column credits of its tiles) and the
// neighbour pairs found there, in original row numbers.  Summing the degrees and concatenating the pairs of all shards
// gives exactly what the un-sharded pass produces; the round loop (graph_rounds) then runs on the assembled graph.
inline void shard_tile_rows(const int64_t tiles, const int shard, const int nShards, unsigned& lo, unsigned& hi) {
  auto bound = [&](const int k) -> unsigned {  // first tile row of shard k: area above it = k / nShards of the triangle
    if (k <= 0) return 0u;
    if (k >= nShards) return static_cast<unsigned>(tiles);
    const double T = static_cast<double>(tiles), total = T * (T + 1.0) / 2.0, want = total * k / nShards;
    // rows 0..m-1 hold m T - m (m - 1) / 2 tiles
    double m = (2.0 * T + 1.0 - std::sqrt((2.0 * T + 1.0) * (2.0 * T + 1.0) - 8.0 * want)) / 2.0;
    return static_cast<unsigned>(std::min<double>(std::max(0.0, std::floor(m + 0.5)), T));
  };
  lo = bound(shard);
  hi = bound(shard + 1);
}

template <int METRIC>
int pairs_impl(const uint32_t* d_x, int64_t N, int fpBits, double cutoff, int shard, int nShards, int32_t* d_counts, int2* d_edges,
               unsigned long long capacity, unsigned long long* h_n_edges, hipStream_t stream) {
  const int   W   = fpBits / 32;
  const float thr = static_cast<float>(1.0 - cutoff);
  const size_t n  = static_cast<size_t>(N);
  StreamScratch tableMem, fp4Mem, sortMem, sortTmp, curMem, tmpCounts;
  CountPlan     plan;
  int           rc = make_plan(plan, tableMem, METRIC, fpBits, thr, stream);
  if (rc != NVMK_OK) return rc;
  const bool sortPass = sorted_first_pass() && METRIC == NVMK_METRIC_TANIMOTO && thr > 0.0f;
  int32_t*   perm     = nullptr;
  NVMK_HIP_CHECK(fp4Mem.alloc(fp4::layout(N, fpBits).bytes, stream));
  if (sortPass) {  // same permutation on every shard: it depends on the fingerprints only
    NVMK_HIP_CHECK(sortMem.alloc(4 * n * sizeof(int32_t), stream));
    int32_t* pop       = sortMem.as<int32_t>();
    int32_t* popSorted = pop + n;
    int32_t* ident     = popSorted + n;
    perm               = ident + n;
    hipLaunchKernelGGL(iota_kernel, dim3(static_cast<unsigned>(ceil_div<int64_t>(N, 256))), dim3(256), 0, stream, ident, N);
    hipLaunchKernelGGL(row_popcount_kernel, dim3(static_cast<unsigned>(ceil_div<int64_t>(N, 256))), dim3(256), 0, stream, d_x, N, W, pop);
    NVMK_LAUNCH_CHECK();
    size_t tmpBytes = 0;
    NVMK_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmpBytes, pop, popSorted, ident, perm, static_cast<int>(N), 0, 32, stream));
    NVMK_HIP_CHECK(sortTmp.alloc(tmpBytes, stream));
    NVMK_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(sortTmp.ptr, tmpBytes, pop, popSorted, ident, perm, static_cast<int>(N), 0, 32, stream));
  }
  rc = fp4::prepare(d_x, perm, N, fpBits, fp4Mem.ptr, stream);
  if (rc != NVMK_OK) return rc;
  const fp4::Prepared PX = fp4::view(fp4Mem.ptr, N, fpBits);
  NVMK_HIP_CHECK(curMem.alloc(256, stream));
  NVMK_HIP_CHECK(hipMemsetAsync(curMem.ptr, 0, 256, stream));
  NVMK_HIP_CHECK(tmpCounts.alloc(n * sizeof(int32_t), stream));
  int32_t* counts = sortPass ? tmpCounts.as<int32_t>() : d_counts;
  NVMK_HIP_CHECK(hipMemsetAsync(counts, 0, n * sizeof(int32_t), stream));
  fp4::CountArgs a{};
  a.metric       = METRIC;
  a.thr          = thr;
  a.table        = plan.table;
  a.tableF       = plan.tableF;
  a.sign         = +1;
  a.nX           = N;
  a.nY           = N;
  a.symmetric    = true;
  a.edges        = d_edges;
  a.edgeCursor   = curMem.as<unsigned long long>();
  a.edgeCapacity = capacity;
  a.bandSkip     = sortPass;
  shard_tile_rows(ceil_div<int64_t>(N, fp4::ROW_PAD), shard, nShards, a.tileRowLo, a.tileRowHi);
  if (a.tileRowHi > a.tileRowLo) {  // an empty band (more shards than tile rows) contributes nothing
    rc = fp4::launch_counts(a, PX, PX, counts, stream);
    if (rc != NVMK_OK) return rc;
  }
  if (sortPass) {
    hipLaunchKernelGGL(scatter_counts_kernel, dim3(static_cast<unsigned>(ceil_div<int64_t>(N, 256))), dim3(256), 0, stream, counts, perm, N,
                       d_counts);
    NVMK_LAUNCH_CHECK();
  }
  unsigned long long nEdges = 0;
  NVMK_HIP_CHECK(hipMemcpyAsync(&nEdges, curMem.ptr, sizeof(nEdges), hipMemcpyDeviceToHost, stream));
  NVMK_HIP_CHECK(hipStreamSynchronize(stream));
  *h_n_edges = nEdges;
  if (nEdges > capacity) {
    set_last_error("butina pairs: %llu neighbour pairs do not fit the buffer of %llu (graph too dense for the sharded path)", nEdges,
                   capacity);
    return NVMK_ERR_OUT_OF_MEMORY;
  }
  if (sortPass && nEdges > 0) {
    const unsigned eb = static_cast<unsigned>(std::min<unsigned long long>(ceil_div<unsigned long long>(nEdges, 256), 65535));
    hipLaunchKernelGGL(remap_edges_kernel, dim3(eb), dim3(256), 0, stream, d_edges, nEdges, perm);
    NVMK_LAUNCH_CHECK();
    NVMK_HIP_CHECK(hipStreamSynchronize(stream));  // scratch (perm) is released at return
  }
  return NVMK_OK;
}

// ===================== dense-matrix Butina =====================================================

// hit = dist <= cutoff (src/butina.cu:1043-1051), written row-major AND transposed so that both
// "row of i" and "column of i" are contiguous (the matrix need not be symmetric).
__global__ __launch_bounds__(NT) void threshold_transpose_kernel(const double* __restrict__ dist,
                                                                 const uint8_t* __restrict__ hitIn,
                                                                 uint8_t* __restrict__ hit, uint8_t* __restrict__ hitT,
                                                                 const int64_t N, const double cutoff) {
  __shared__ uint8_t tile[64][65];
  const int64_t      r0 = static_cast<int64_t>(blockIdx.y) * 64;
  const int64_t      c0 = static_cast<int64_t>(blockIdx.x) * 64;
  const int          tx = threadIdx.x & 63;
  const int          ty = threadIdx.x >> 6;
  for (int rr = ty; rr < 64; rr += NT / 64) {
    const int64_t r = r0 + rr, c = c0 + tx;
    uint8_t       h = 0;
    if (r < N && c < N) {
      h              = dist ? static_cast<uint8_t>(dist[r * N + c] <= cutoff) : static_cast<uint8_t>(hitIn[r * N + c] != 0);
      hit[r * N + c] = h;
    }
    tile[rr][tx] = h;
  }
  __syncthreads();
  for (int cc = ty; cc < 64; cc += NT / 64) {
    const int64_t c = c0 + cc, r = r0 + tx;
    if (r < N && c < N) hitT[c * N + r] = tile[tx][cc];
  }
}

// counts[i] = #{j : hit[i][j]} (all points unassigned at the start), one wave per row.
__global__ __launch_bounds__(NT) void dense_degree_kernel(const uint8_t* __restrict__ hit, const int64_t N,
                                                          int32_t* __restrict__ counts) {
  const int64_t r    = static_cast<int64_t>(blockIdx.x) * (NT / 64) + (threadIdx.x >> 6);
  const int     lane = threadIdx.x & 63;
  if (r >= N) return;
  int n = 0;
  for (int64_t j = lane; j < N; j += 64) n += hit[r * N + j] ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
  if (lane == 0) counts[r] = n;
}

struct DenseState {
  unsigned long long bestKey[2];
  int32_t            nClusters;
  int32_t            nMembers;
  int32_t            done;
  int32_t            pad;
};

// argmax over unassigned points with ties toward the highest index (lastArgMaxKernel, src/butina.cu:464-481).
__global__ __launch_bounds__(NT) void dense_argmax_kernel(DenseState* __restrict__ st, const int32_t* __restrict__ counts,
                                                          const int32_t* __restrict__ clusters, const int64_t N,
                                                          const int parity) {
  if (st->done) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) st->bestKey[parity ^ 1] = 0ull;
  unsigned long long best = 0ull;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * NT + threadIdx.x; i < N; i += static_cast<int64_t>(gridDim.x) * NT) {
    if (clusters[i] < 0) {
      const unsigned long long key =
        (static_cast<unsigned long long>(static_cast<unsigned>(counts[i])) << 32) | static_cast<unsigned>(i);
      best = key > best ? key : best;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(best, o);
    best                           = other > best ? other : best;
  }
  __shared__ unsigned long long wbest[NT / 64];
  if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < NT / 64; ++w) best = wbest[w] > best ? wbest[w] : best;
    if (best != 0ull) atomicMax(&st->bestKey[parity], best);
  }
}

// Assign the centroid's unassigned neighbours (src/butina.cu:241-269) and list them in `members`.
__global__ __launch_bounds__(NT) void dense_assign_kernel(DenseState* __restrict__ st, const uint8_t* __restrict__ hit,
                                                          int32_t* __restrict__ clusters, int32_t* __restrict__ members,
                                                          int32_t* __restrict__ centroids, const int64_t N,
                                                          const int parity) {
  if (st->done) return;
  const unsigned long long key = st->bestKey[parity];
  const int                sz  = static_cast<int>(key >> 32);
  if (sz < 2) return;  // kMinLoopSizeForAssignment (src/butina.cu:34); dense_finish marks done
  const int64_t c   = static_cast<int64_t>(key & 0xffffffffull);
  const int     cid = st->nClusters;
  const int64_t i   = static_cast<int64_t>(blockIdx.x) * NT + threadIdx.x;
  if (i < N && clusters[i] < 0 && (hit[c * N + i] || i == c)) {
    clusters[i]                           = cid;
    members[atomicAdd(&st->nMembers, 1)] = static_cast<int32_t>(i);
  }
  if (i == 0) centroids[cid] = static_cast<int32_t>(c);
}

// counts[i] -= #{m in members : hit[i][m]} for unassigned i, using the transposed matrix so that a
// member's column is contiguous.  grid.y strides the members.
__global__ __launch_bounds__(NT) void dense_subtract_kernel(const DenseState* __restrict__ st,
                                                            const uint8_t* __restrict__ hitT,
                                                            const int32_t* __restrict__ clusters,
                                                            const int32_t* __restrict__ members,
                                                            int32_t* __restrict__ counts, const int64_t N,
                                                            const int parity) {
  if (st->done) return;
  if (static_cast<int>(st->bestKey[parity] >> 32) < 2) return;
  const int     nM = st->nMembers;
  const int64_t i  = static_cast<int64_t>(blockIdx.x) * NT + threadIdx.x;
  if (i >= N || clusters[i] >= 0) return;
  int dec = 0;
  for (int m = blockIdx.y; m < nM; m += gridDim.y) {
    dec += hitT[static_cast<int64_t>(members[m]) * N + i] ? 1 : 0;
  }
  if (dec) atomicSub(&counts[i], dec);
}

__global__ void dense_finish_kernel(DenseState* __restrict__ st, const int parity) {
  if (st->done) return;
  if (static_cast<int>(st->bestKey[parity] >> 32) < 2) {
    st->done = 1;
    return;
  }
  st->nClusters += 1;
  st->nMembers = 0;
}

// ---- singleton ids and renumbering by size, on the device (src/butina.cu:281-307, :369-448) -----------------------
// flags[i] = 1 for points no cluster took; their exclusive prefix sum ranks them in ascending index order.
__global__ void dense_unassigned_flags_kernel(const int32_t* __restrict__ clusters, const int64_t N, int32_t* __restrict__ flags) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < N) flags[i] = clusters[i] < 0 ? 1 : 0;
}
// singleton cluster ids follow the greedy clusters in ascending point order; sizes are histogrammed on the way
__global__ void dense_singletons_kernel(const DenseState* __restrict__ st, int32_t* __restrict__ clusters, const int32_t* __restrict__ rank,
                                        const int64_t N, int32_t* __restrict__ centroids, int32_t* __restrict__ sizes) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int c = clusters[i];
  if (c < 0) {
    c            = st->nClusters + rank[i];
    clusters[i]  = c;
    centroids[c] = static_cast<int32_t>(i);
  }
  atomicAdd(&sizes[c], 1);
}
__global__ void dense_total_kernel(const DenseState* __restrict__ st, const int32_t* __restrict__ rank, const int32_t* __restrict__ flags,
                                   const int64_t N, int32_t* __restrict__ nTotal) {
  *nTotal = st->nClusters + rank[N - 1] + flags[N - 1];
}
// order[newId] = old id (ids sorted by size, descending, stable) -> remap[old] = newId, newCent[newId] = cent[old]
__global__ void dense_remap_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ nTotal, const int32_t* __restrict__ cent,
                                   int32_t* __restrict__ remap, int32_t* __restrict__ newCent) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= *nTotal) return;
  const int old = order[k];
  remap[old]    = k;
  if (newCent) newCent[k] = cent[old];
}
__global__ void dense_apply_remap_kernel(int32_t* __restrict__ clusters, const int32_t* __restrict__ remap, const int64_t N) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < N) clusters[i] = remap[clusters[i]];
}
__global__ void iota32_kernel(int32_t* __restrict__ v, const int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) v[i] = static_cast<int32_t>(i);
}

}  // namespace butina
}  // namespace nvmk

using namespace nvmk;
using namespace nvmk::butina;

extern "C" {

int nvmk_neighbor_counts(int metric, const uint32_t* d_x, const int32_t* d_x_rows, int64_t nX, const uint32_t* d_y,
                         const int32_t* d_y_rows, int64_t nY, int fp_bits, float threshold, int sign,
                         int32_t* d_counts, void* stream) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(sign == 1 || sign == -1, "neighbor counts: sign must be +1 or -1, got %d", sign);
  NVMK_REQUIRE(nX >= 0 && nY >= 0, "neighbor counts: negative row count");
  if (nX == 0 || nY == 0) return NVMK_OK;
  NVMK_REQUIRE(d_x && d_y && d_counts, "neighbor counts: NULL buffer");
  StreamScratch tableMem;
  CountPlan     plan;
  const int     rc = make_plan(plan, tableMem, metric, fp_bits, threshold, as_stream(stream));
  if (rc != NVMK_OK) return rc;
  hipStream_t  s       = as_stream(stream);
  const double pairs   = static_cast<double>(nX) * static_cast<double>(nY);
  const bool   useMfma = !force_valu() && (force_mfma() || (pairs >= 4.0e6 && nX >= 64 && nY >= 64));
  if (!useMfma) {
    return launch_counts(plan, d_x, d_x_rows, nX, nullptr, d_y, d_y_rows, nY, nullptr, sign, d_counts, s);
  }
  // gather + expand both operands into compact prepared sets; counts stay indexed by the caller's row ids
  StreamScratch wsX, wsY;
  NVMK_HIP_CHECK(wsX.alloc(fp4::layout(nX, fp_bits).bytes, s));
  int rc2 = fp4::prepare(d_x, d_x_rows, nX, fp_bits, wsX.ptr, s);
  if (rc2 != NVMK_OK) return rc2;
  if (d_x == d_y && nX == nY && d_x_rows == nullptr && d_y_rows == nullptr) {
    // a set against ITSELF: the similarity is symmetric, so only the tiles on or above the diagonal are evaluated and every pair
    // credits both of its rows — half the pairs for the same counts (self pairs included, as in the rectangular pass)
    fp4::CountArgs a{};
    a.metric    = metric;
    a.thr       = threshold;
    a.table     = plan.table;
    a.tableF    = plan.tableF;
    a.sign      = sign;
    a.nX        = nX;
    a.nY        = nX;
    a.symmetric = true;
    const fp4::Prepared P = fp4::view(wsX.ptr, nX, fp_bits);
    return fp4::launch_counts(a, P, P, d_counts, s);
  }
  NVMK_HIP_CHECK(wsY.alloc(fp4::layout(nY, fp_bits).bytes, s));
  rc2 = fp4::prepare(d_y, d_y_rows, nY, fp_bits, wsY.ptr, s);
  if (rc2 != NVMK_OK) return rc2;
  fp4::CountArgs a{};
  a.metric = metric;
  a.thr    = threshold;
  a.table  = plan.table;
  a.tableF = plan.tableF;
  a.sign   = sign;
  a.xIds   = d_x_rows;
  a.nX     = nX;
  a.nY     = nY;
  return fp4::launch_counts(a, fp4::view(wsX.ptr, nX, fp_bits), fp4::view(wsY.ptr, nY, fp_bits), d_counts, s);
}

int nvmk_butina_fused(int metric, const uint32_t* d_x, int64_t N, int fp_bits, double cutoff,
                      int32_t* h_cluster_indices, int64_t* h_offsets, int32_t* h_centroids, int64_t* n_clusters,
                      void* stream) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(metric == NVMK_METRIC_TANIMOTO || metric == NVMK_METRIC_COSINE, "unknown metric %d", metric);
  NVMK_REQUIRE(cutoff >= 0.0 && cutoff <= 1.0, "cutoff must be in [0, 1], got %g", cutoff);
  NVMK_REQUIRE(fp_bits > 0 && fp_bits % 32 == 0, "fp_bits must be a positive multiple of 32, got %d", fp_bits);
  NVMK_REQUIRE(N >= 0, "negative N");
  NVMK_REQUIRE(h_offsets && n_clusters, "NULL output");
  h_offsets[0] = 0;
  *n_clusters  = 0;
  if (N == 0) return NVMK_OK;
  NVMK_REQUIRE(d_x && h_cluster_indices && h_centroids, "NULL buffer");
  return metric == NVMK_METRIC_TANIMOTO ?
           fused_impl<NVMK_METRIC_TANIMOTO>(d_x, N, fp_bits, cutoff, h_cluster_indices, h_offsets, h_centroids,
                                            n_clusters, as_stream(stream)) :
           fused_impl<NVMK_METRIC_COSINE>(d_x, N, fp_bits, cutoff, h_cluster_indices, h_offsets, h_centroids,
                                          n_clusters, as_stream(stream));
}

int nvmk_butina_pairs(int metric, const uint32_t* d_x, int64_t N, int fp_bits, double cutoff, int shard, int n_shards,
                      int32_t* d_counts, int32_t* d_pairs, uint64_t pair_capacity, uint64_t* h_n_pairs, void* stream) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(metric == NVMK_METRIC_TANIMOTO || metric == NVMK_METRIC_COSINE, "unknown metric %d", metric);
  NVMK_REQUIRE(cutoff >= 0.0 && cutoff <= 1.0, "cutoff must be in [0, 1], got %g", cutoff);
  NVMK_REQUIRE(fp_bits > 0 && fp_bits % 128 == 0, "fp_bits must be a positive multiple of 128, got %d", fp_bits);
  NVMK_REQUIRE(n_shards >= 1 && shard >= 0 && shard < n_shards, "butina pairs: bad shard %d of %d", shard, n_shards);
  NVMK_REQUIRE(N >= 0 && N <= 0x7fffffffLL, "butina pairs: bad N %lld", (long long)N);
  NVMK_REQUIRE(h_n_pairs != nullptr, "butina pairs: NULL output");
  *h_n_pairs = 0;
  if (N == 0) return NVMK_OK;
  NVMK_REQUIRE(d_x && d_counts && (d_pairs || pair_capacity == 0), "butina pairs: NULL buffer");
  NVMK_REQUIRE(reinterpret_cast<uintptr_t>(d_x) % 16 == 0, "butina pairs: the fingerprint matrix must be 16-byte aligned");
  unsigned long long nE = 0;
  const int rc = metric == NVMK_METRIC_TANIMOTO ?
                   pairs_impl<NVMK_METRIC_TANIMOTO>(d_x, N, fp_bits, cutoff, shard, n_shards, d_counts, reinterpret_cast<int2*>(d_pairs),
                                                    pair_capacity, &nE, as_stream(stream)) :
                   pairs_impl<NVMK_METRIC_COSINE>(d_x, N, fp_bits, cutoff, shard, n_shards, d_counts, reinterpret_cast<int2*>(d_pairs),
                                                  pair_capacity, &nE, as_stream(stream));
  *h_n_pairs = nE;
  return rc;
}

}  // extern "C"

namespace {
// caller-supplied neighbour graph: every index inside [0, N), no self pairs, and the degree vector what the pairs imply
// (1 + pairs of the row; 0 for a row without any bit set, which has no pair either) — the CSR fill trusts both
__global__ void check_pairs_kernel(const int2* __restrict__ pairs, const unsigned long long nPairs, const int32_t n,
                                   int32_t* __restrict__ deg, int32_t* __restrict__ bad) {
  for (unsigned long long e = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < nPairs;
       e += static_cast<unsigned long long>(gridDim.x) * blockDim.x) {
    const int2 p = pairs[e];
    if (p.x < 0 || p.x >= n || p.y < 0 || p.y >= n || p.x == p.y) {
      atomicOr(bad, 1);
    } else {
      atomicAdd(&deg[p.x], 1);
      atomicAdd(&deg[p.y], 1);
    }
  }
}
__global__ void check_degrees_kernel(const int32_t* __restrict__ counts, const int32_t* __restrict__ deg, const int32_t n,
                                     int32_t* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && counts[i] != deg[i] + 1 && !(counts[i] == 0 && deg[i] == 0)) atomicOr(bad, 2);
}
}  // namespace

extern "C" {

int nvmk_butina_from_pairs(int64_t N, const int32_t* d_counts, const int32_t* d_pairs, uint64_t n_pairs, int32_t* h_cluster_indices,
                           int64_t* h_offsets, int32_t* h_centroids, int64_t* n_clusters, void* stream_) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(N >= 0 && N <= 0x7fffffffLL, "butina from pairs: bad N %lld", (long long)N);
  NVMK_REQUIRE(h_offsets && n_clusters, "NULL output");
  h_offsets[0] = 0;
  *n_clusters  = 0;
  if (N == 0) return NVMK_OK;
  NVMK_REQUIRE(d_counts && h_cluster_indices && h_centroids && (d_pairs || n_pairs == 0), "butina from pairs: NULL buffer");
  hipStream_t   stream = as_stream(stream_);
  {
    StreamScratch chk;
    NVMK_HIP_CHECK(chk.alloc((static_cast<size_t>(N) + 1) * sizeof(int32_t), stream));
    NVMK_HIP_CHECK(hipMemsetAsync(chk.ptr, 0, (static_cast<size_t>(N) + 1) * sizeof(int32_t), stream));
    int32_t* deg = chk.as<int32_t>();
    int32_t* bad = deg + N;
    if (n_pairs > 0) {
      const unsigned eb = static_cast<unsigned>(std::min<unsigned long long>(ceil_div<unsigned long long>(n_pairs, 256), 65535));
      hipLaunchKernelGGL(check_pairs_kernel, dim3(eb), dim3(256), 0, stream, reinterpret_cast<const int2*>(d_pairs), n_pairs,
                         static_cast<int32_t>(N), deg, bad);
    }
    hipLaunchKernelGGL(check_degrees_kernel, dim3(static_cast<unsigned>(ceil_div<int64_t>(N, 256))), dim3(256), 0, stream, d_counts, deg,
                       static_cast<int32_t>(N), bad);
    NVMK_LAUNCH_CHECK();
    int32_t hBad = 0;
    NVMK_HIP_CHECK(hipMemcpyAsync(&hBad, bad, sizeof(hBad), hipMemcpyDeviceToHost, stream));
    NVMK_HIP_CHECK(hipStreamSynchronize(stream));
    NVMK_REQUIRE((hBad & 1) == 0, "butina from pairs: a pair names a row outside [0, %lld) or a row twice", (long long)N);
    NVMK_REQUIRE((hBad & 2) == 0, "butina from pairs: counts is not the degree vector of the pair list (1 + pairs of a row)");
  }
  StreamScratch mem;
  RoundBuffers  rb{};
  int           rc = alloc_round_buffers(mem, N, stream, rb);
  if (rc != NVMK_OK) return rc;
  NVMK_HIP_CHECK(hipMemcpyAsync(rb.counts, d_counts, static_cast<size_t>(N) * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
  LoopState snap{};
  rc = graph_rounds(rb, N, reinterpret_cast<const int2*>(d_pairs), n_pairs, stream, snap);
  if (rc != NVMK_OK) return rc;
  return read_back(rb, N, snap, h_cluster_indices, h_offsets, h_centroids, n_clusters, stream);
}

int nvmk_butina_dense(const double* d_dist, const uint8_t* d_hit, int64_t N, double cutoff, int neighborlist_max_size,
                      int32_t* d_clusters, int32_t* d_centroids, int64_t* h_n_clusters, void* stream_) {
  NVMK_MARK_ENTRY();
  // neighborlist_max_size only tunes the reference's small-cluster phase (src/butina.cu:975-1004: clusters below that
  // size are resolved from per-point neighbour lists instead of matrix rows).  This implementation has a single phase —
  // the hit matrix is thresholded to bytes once and every round reads contiguous rows / columns of it — so the parameter
  // changes nothing here: it is validated for API compatibility (nvmolkit/clustering.py:79-82) and IGNORED
  // (include/nvmolkit_amd.h says so).
  const int nl = neighborlist_max_size;
  NVMK_REQUIRE(nl == 8 || nl == 16 || nl == 24 || nl == 32 || nl == 64 || nl == 128,
               "neighborlistMaxSize must be 8, 16, 24, 32, 64, or 128. Got: %d", nl);
  NVMK_REQUIRE(N >= 0 && N <= 0x7fffffffLL, "butina: bad N %lld", (long long)N);
  if (h_n_clusters) *h_n_clusters = 0;
  if (N == 0) return NVMK_OK;
  NVMK_REQUIRE((d_dist != nullptr) != (d_hit != nullptr), "butina: pass exactly one of d_dist / d_hit");
  NVMK_REQUIRE(d_clusters != nullptr, "butina: d_clusters is NULL");
  hipStream_t  stream = as_stream(stream_);
  const size_t n      = static_cast<size_t>(N);

  StreamScratch matMem, mem;
  NVMK_HIP_CHECK(matMem.alloc(2 * n * n, stream));
  uint8_t* hit  = matMem.as<uint8_t>();
  uint8_t* hitT = hit + n * n;
  NVMK_HIP_CHECK(mem.alloc((16 + 3 * n) * sizeof(int32_t), stream));
  auto*    st        = mem.as<DenseState>();
  int32_t* counts    = mem.as<int32_t>() + 16;
  int32_t* members   = counts + n;
  int32_t* centroids = members + n;
  NVMK_HIP_CHECK(hipMemsetAsync(st, 0, 16 * sizeof(int32_t), stream));
  NVMK_HIP_CHECK(hipMemsetAsync(d_clusters, 0xff, n * sizeof(int32_t), stream));  // -1

  const unsigned t64 = static_cast<unsigned>(ceil_div<int64_t>(N, 64));
  NVMK_REQUIRE(t64 <= 65535, "butina: N too large for the dense path (%lld)", (long long)N);
  hipLaunchKernelGGL(threshold_transpose_kernel, dim3(t64, t64), dim3(NT), 0, stream, d_dist, d_hit, hit, hitT, N, cutoff);
  hipLaunchKernelGGL(dense_degree_kernel, dim3(static_cast<unsigned>(ceil_div<int64_t>(N, NT / 64))), dim3(NT), 0, stream,
                     hit, N, counts);
  NVMK_LAUNCH_CHECK();

  const unsigned nb = static_cast<unsigned>(ceil_div<int64_t>(N, NT));
  const unsigned ab = std::min(nb, 1024u);
  DenseState     snap{};
  int64_t        round = 0;
  for (;;) {
    for (int b = 0; b < 32; ++b, ++round) {
      const int parity = static_cast<int>(round & 1);
      hipLaunchKernelGGL(dense_argmax_kernel, dim3(ab), dim3(NT), 0, stream, st, counts, d_clusters, N, parity);
      hipLaunchKernelGGL(dense_assign_kernel, dim3(nb), dim3(NT), 0, stream, st, hit, d_clusters, members, centroids, N,
                         parity);
      hipLaunchKernelGGL(dense_subtract_kernel, dim3(nb, 16), dim3(NT), 0, stream, st, hitT, d_clusters, members, counts,
                         N, parity);
      hipLaunchKernelGGL(dense_finish_kernel, dim3(1), dim3(1), 0, stream, st, parity);
    }
    NVMK_LAUNCH_CHECK();
    NVMK_HIP_CHECK(hipMemcpyAsync(&snap, st, sizeof(snap), hipMemcpyDeviceToHost, stream));
    NVMK_HIP_CHECK(hipStreamSynchronize(stream));
    if (snap.done) break;
  }

  // singletons (ascending index) + renumber by descending size, stable by original id (assignSingletonIdsKernel
  // src/butina.cu:281-307, renumberClustersBySize :369-448) — on the device: prefix sum of the unassigned flags, size
  // histogram, one stable radix sort of the cluster ids by size, two remap kernels.  Only the cluster count comes back.
  StreamScratch rn, tmp;
  NVMK_HIP_CHECK(rn.alloc((8 * n + 16) * sizeof(int32_t), stream));
  int32_t* flags   = rn.as<int32_t>();
  int32_t* rank    = flags + n;
  int32_t* sizes   = rank + n;
  int32_t* sizesS  = sizes + n;
  int32_t* ids     = sizesS + n;
  int32_t* order   = ids + n;
  int32_t* remap   = order + n;
  int32_t* newCent = remap + n;
  int32_t* nTotal  = newCent + n;
  const unsigned nb256 = static_cast<unsigned>(ceil_div<int64_t>(N, 256));
  NVMK_HIP_CHECK(hipMemsetAsync(sizes, 0, n * sizeof(int32_t), stream));
  hipLaunchKernelGGL(dense_unassigned_flags_kernel, dim3(nb256), dim3(256), 0, stream, d_clusters, N, flags);
  size_t tmpBytes = 0, sortBytes = 0;
  NVMK_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, flags, rank, static_cast<int>(N), stream));
  NVMK_HIP_CHECK(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, sortBytes, sizes, sizesS, ids, order, static_cast<int>(N), 0, 32, stream));
  NVMK_HIP_CHECK(tmp.alloc(std::max(tmpBytes, sortBytes), stream));
  NVMK_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.ptr, tmpBytes, flags, rank, static_cast<int>(N), stream));
  hipLaunchKernelGGL(dense_total_kernel, dim3(1), dim3(1), 0, stream, st, rank, flags, N, nTotal);
  hipLaunchKernelGGL(dense_singletons_kernel, dim3(nb256), dim3(256), 0, stream, st, d_clusters, rank, N, centroids, sizes);
  hipLaunchKernelGGL(iota32_kernel, dim3(nb256), dim3(256), 0, stream, ids, N);
  NVMK_LAUNCH_CHECK();
  int32_t nTot = 0;
  NVMK_HIP_CHECK(hipMemcpyAsync(&nTot, nTotal, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  NVMK_HIP_CHECK(hipStreamSynchronize(stream));
  const int64_t nC = nTot;
  // ids beyond nC have size 0 and sort behind every real cluster (stable), so sorting all N slots is harmless
  NVMK_HIP_CHECK(hipcub::DeviceRadixSort::SortPairsDescending(tmp.ptr, sortBytes, sizes, sizesS, ids, order, static_cast<int>(nC), 0, 32, stream));
  const unsigned cb = static_cast<unsigned>(ceil_div<int64_t>(std::max<int64_t>(nC, 1), 256));
  hipLaunchKernelGGL(dense_remap_kernel, dim3(cb), dim3(256), 0, stream, order, nTotal, centroids, remap, d_centroids ? newCent : nullptr);
  hipLaunchKernelGGL(dense_apply_remap_kernel, dim3(nb256), dim3(256), 0, stream, d_clusters, remap, N);
  NVMK_LAUNCH_CHECK();
  if (d_centroids) {
    NVMK_HIP_CHECK(hipMemcpyAsync(d_centroids, newCent, static_cast<size_t>(nC) * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
  }
  NVMK_HIP_CHECK(hipStreamSynchronize(stream));  // scratch is released in stream order; the caller reads d_clusters next
  if (h_n_clusters) *h_n_clusters = nC;
  return NVMK_OK;
}

}  // extern "C"
