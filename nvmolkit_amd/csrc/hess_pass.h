// The inverse-Hessian pass of the fused BFGS kernels (one workgroup of 256 threads per system) — shared by
// minimize.hip and tools/ubench_hess.hip (the pass is the hot spot of the conformer path; the microbenchmark runs it alone).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace nvmk {
namespace minim {

constexpr int NT = 256;

// ---- inverse-Hessian pass -------------------------------------------------------------------------
// The inverse Hessian is symmetric: only its lower triangle is stored, row r holding columns 0..r padded to an even
// length (rows stay 16-byte aligned, the pad entry stays 0).  ONE pass per BFGS iteration streams it once with the
// whole workgroup: it applies the rank-2 update that the PREVIOUS iteration left pending
//     H += rfac xi xi^T - fad hdg hdg^T + fae u u^T          (RDKit BFGSOpt.h; u = rfac xi - fad hdg)
// writes the element back, and accumulates t = H g for the current gradient.  The two products the textbook loop
// needs per iteration follow from t without touching the matrix again:
//     H dGrad       = H g_new - H g_old = t - hg                       (hg = H g of the previous iterate, kept in LDS)
//     H_new g_new   = t + rfac xi (xi.g) - fad hdg (hdg.g) + fae u (u.g)
// so the traffic per iteration is one read + one write of n (n + 2) / 2 doubles instead of two reads + one write of n^2.
//
// Work split (round 2): ONE WAVE PER ROW.  Wave w owns rows w, w + NW, ...; its 64 lanes span 128 consecutive columns of
// the row as 16-byte pairs, so every load / store of a row is one fully coalesced (global) or conflict-free (LDS) wave
// instruction.  A lane keeps the vector entries of ITS two columns (xi, hdg, u, g) in registers for the whole column
// chunk; the row's coefficients are wave-uniform.  Row sums finish with a DPP reduction inside the wave (no LDS), mirrored
// (column) sums stay in the lane's registers over all the rows of the chunk and are written once per wave.  Every partial
// sum has a single writer and the final sum runs in a fixed order: a minimisation is reproducible run to run.
// (Round 1 split rows over 8 row groups x 32 lanes with per-group partial-sum slabs in LDS: measured 20-27 us per pass at
// n = 144-192 whether the matrix came from HBM or from LDS — the pass was bound by its own chain of LDS read-modify-writes
// and half-wave shuffles, not by bandwidth: profiles/r02_conformers/.)
constexpr int NW = NT / 64;  // waves per workgroup

__host__ __device__ __forceinline__ int64_t hess_row_offset(const int64_t r) {  // rows 0..r-1, each padded to even length
  return ((r + 1) >> 1) * ((r | 1) + 1);  // r even: r (r + 2) / 2, r odd: (r + 1)^2 / 2 — branch-free
}

// Rows [0, Rl) of the packed triangle live in LDS behind the vectors (as many as the launch's LDS budget holds), rows Rl..
// stream from HBM.
__host__ __device__ __forceinline__ int resident_rows(const int n, const int64_t hldsDoubles) {
  if (hess_row_offset(n) <= hldsDoubles) return n;
  int r = 0;
  while (r < n && hess_row_offset(r + 1) <= hldsDoubles) ++r;
  return r;
}
// LDS layout of bfgs_kernel: 11 vectors + (1 + NW) partial-sum slabs of n doubles (row sums, then one slab of mirrored-entry
// sums per wave; the per-wave gradient slabs alias them), 16 doubles of reduction scratch, then the resident rows of the
// inverse Hessian in whatever the launch's dynamic LDS (ldsDoubles) leaves.
__host__ __device__ constexpr int64_t lds_vector_doubles(const int64_t n) { return (12 + NW) * n + 16; }
__host__ __device__ constexpr int64_t lds_hessian_doubles(const int64_t ldsDoubles, const int64_t n) {
  return ldsDoubles > lds_vector_doubles(n) ? ldsDoubles - lds_vector_doubles(n) : 0;
}

// Sum over the 64 lanes of a wave, result in every lane.  DPP moves only (no LDS traffic): butterflies inside a quad and
// a row of 16, then the two row broadcasts of gfx9; the order of the additions is fixed.
template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ double dpp_mov(const double x) {
  if constexpr (ROW_MASK == 0xf) {  // every lane has a source (permutations inside a quad / a row): no "old" value needed
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  } else {  // row broadcasts: rows outside the mask add 0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
  }
}
// N independent sums at once, step by step: the N dependency chains interleave instead of running one after the other.
template <int N> __device__ __forceinline__ void wave_sum_n(double (&v)[N]) {
#pragma unroll
  for (int u = 0; u < N; ++u) v[u] += dpp_mov<0xb1>(v[u]);  // quad_perm [1, 0, 3, 2]
#pragma unroll
  for (int u = 0; u < N; ++u) v[u] += dpp_mov<0x4e>(v[u]);  // quad_perm [2, 3, 0, 1]
#pragma unroll
  for (int u = 0; u < N; ++u) v[u] += dpp_mov<0x124>(v[u]);  // row_ror 4
#pragma unroll
  for (int u = 0; u < N; ++u) v[u] += dpp_mov<0x128>(v[u]);  // row_ror 8: every lane of a row of 16 holds the row's sum
#pragma unroll
  for (int u = 0; u < N; ++u) v[u] += dpp_mov<0x142, 0xa>(v[u]);  // row_bcast 15 into rows 1 and 3
#pragma unroll
  for (int u = 0; u < N; ++u) v[u] += dpp_mov<0x143, 0xc>(v[u]);  // row_bcast 31 into rows 2 and 3: lane 63 holds the total
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v[u]), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v[u]), 63);
    v[u]         = __hiloint2double(hi, lo);
  }
}

// Four row sums at once ("transposed" reduction): two merge steps inside a quad leave lane l with the quad's sum of row
// (l & 3); rotations inside the row of 16 and two cross-row exchanges finish it.  Every lane ends up with the total of row
// (lane & 3): ~9 instructions per row instead of ~20 for four separate wave reductions, and no scalar read-back.
__device__ __forceinline__ double wave_sum4_transposed(const double (&rs)[4], const int lane) {
  const bool odd = (lane & 1) != 0, up = (lane & 2) != 0;
  const double ab = (odd ? rs[1] : rs[0]) + dpp_mov<0xb1>(odd ? rs[0] : rs[1]);  // even lanes: row 0 pair sum, odd lanes: row 1
  const double cd = (odd ? rs[3] : rs[2]) + dpp_mov<0xb1>(odd ? rs[2] : rs[3]);  // even lanes: row 2,          odd lanes: row 3
  double       x  = (up ? cd : ab) + dpp_mov<0x4e>(up ? ab : cd);                // lane & 3 = row: sum over the quad
  x += dpp_mov<0x124>(x);  // row_ror 4  (keeps lane & 3)
  x += dpp_mov<0x128>(x);  // row_ror 8: sum over the row of 16 lanes
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}

// State of one column chunk of 128 columns: this lane's two columns and the vector entries that belong to them.
struct HessChunk {
  int    c0;
  double g0, g1, x0, x1, h0, h1, u0, u1;
};

// Rows [rFrom, rEnd) of NCH adjacent column chunks (128 NCH columns); the matrix row `rBase` starts at H (LDS or HBM: the
// address space is known at the call).  A wave owns rows w, w + NW, ...; four of its rows form a group whose loads are all
// requested before the first use and whose four row sums are reduced together (wave_sum4_transposed); with PREFETCH (HBM
// rows) the NEXT group's matrix pairs are requested before the current group is worked on — ahead of the current group's
// stores, so waiting for them does not wait for the stores (vmcnt is in-order).  Per-row overhead (coefficients, reduction,
// row-sum update) is shared by the NCH chunks: the pass is bound by instruction issue, not by bandwidth, at these sizes.
// The wave index is scalar: row numbers, row offsets and the branches on them live on the scalar unit.
// col[k][0..1] (mirrored-entry sums of the lane's columns) are carried by the caller across the LDS and the HBM range, so
// every sum is formed in the same order wherever the rows live: results do not depend on the residency split.
template <int NCH, bool PREFETCH>
__device__ __forceinline__ void hess_range(double* __restrict__ H, const int rBase, const int rFrom, const int rEnd, const int wave,
                                           const int lane, const HessChunk (&ck)[NCH], const bool pending, const double rfac,
                                           const double fad, const double fae, const double* __restrict__ xi,
                                           const double* __restrict__ hdg, const double* __restrict__ uu,
                                           const double* __restrict__ g, double* __restrict__ rowsum, double (&col)[NCH][2]) {
  constexpr int RU   = 4;
  const int     base = static_cast<int>(hess_row_offset(rBase));
  auto row_ptr = [&](const int r) -> double* { return H + (static_cast<int>(hess_row_offset(r)) - base); };
  auto load_group = [&](const int r0, double2 (&dst)[RU][NCH]) {
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int r = r0 + NW * u;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        dst[u][k] = make_double2(0.0, 0.0);
        if (r < rEnd && ck[k].c0 <= r) dst[u][k] = *reinterpret_cast<const double2*>(row_ptr(r) + ck[k].c0);
      }
    }
  };
  int     r0 = rFrom + ((wave - rFrom) % NW + NW) % NW;  // first row of this wave at or after rFrom
  double2 next[RU][NCH];
  if constexpr (PREFETCH) load_group(r0, next);
  for (; r0 < rEnd; r0 += NW * RU) {
    double2 hv[RU][NCH];
    if constexpr (PREFETCH) {
#pragma unroll
      for (int u = 0; u < RU; ++u) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) hv[u][k] = next[u][k];
      }
      load_group(r0 + NW * RU, next);  // rows past rEnd load nothing
    } else {
      load_group(r0, hv);
    }
    // the row this LANE will write the sum of (lane & 3 selects it, see wave_sum4_transposed) and its running sum
    const int    myRow = r0 + NW * (lane & 3);
    const double rold  = (lane < 4 && myRow < rEnd) ? rowsum[myRow] : 0.0;
    double       gr[RU], ar[RU], br[RU], dr[RU], rs[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int r  = r0 + NW * u;
      const int rc = r < rEnd ? r : rFrom;
      gr[u]        = g[rc];
      ar[u]        = pending ? rfac * xi[rc] : 0.0;
      br[u]        = pending ? fad * hdg[rc] : 0.0;
      dr[u]        = pending ? fae * uu[rc] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int r = r0 + NW * u;
      rs[u]       = 0.0;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int c0 = ck[k].c0;
        double2   h  = hv[u][k];
        if (r < rEnd && pending && c0 <= r) {
          h.x += ar[u] * ck[k].x0 - br[u] * ck[k].h0 + dr[u] * ck[k].u0;
          if (c0 + 1 <= r) h.y += ar[u] * ck[k].x1 - br[u] * ck[k].h1 + dr[u] * ck[k].u1;  // the pad entry stays 0
          *reinterpret_cast<double2*>(row_ptr(r) + c0) = h;
        }
        if (r < rEnd && c0 < r) col[k][0] += h.x * gr[u];  // mirrored entries (strictly below the diagonal)
        if (r < rEnd && c0 + 1 < r) col[k][1] += h.y * gr[u];
        rs[u] += h.x * ck[k].g0 + h.y * ck[k].g1;  // lanes past the row (and rows past the range) hold zeros
      }
    }
    const double tot = wave_sum4_transposed(rs, lane);
    if (lane < 4 && myRow < rEnd) rowsum[myRow] = rold + tot;  // one writer per row (this wave), column super-chunks in order
  }
}

// The pass: rows [0, Rl) from LDS (Hl), rows [Rl, n) from HBM (Hg, whose first element is row Rl's).  `part` = row sums
// [n] (zero on entry, visible to the workgroup) then NW slabs [n] of mirrored-entry sums (written here).
template <int NCH, bool PREFETCH>
__device__ __forceinline__ void hess_pass_chunks(double* __restrict__ Hl, double* __restrict__ Hg, const int Rl, const int n,
                                                 const bool pending, const double rfac, const double fad, const double fae,
                                                 const double* __restrict__ xi, const double* __restrict__ hdg,
                                                 const double* __restrict__ uu, const double* __restrict__ g,
                                                 double* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  double*   rowsum = part;
  double*   colsum = part + (1 + wave) * n;
  for (int cBase = 0; cBase < n; cBase += 128 * NCH) {  // column super-chunk: rows before cBase have no column in it
    HessChunk ck[NCH];
    double    col[NCH][2];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int  c0  = cBase + 128 * k + 2 * lane;
      const bool in0 = c0 < n, in1 = c0 + 1 < n;
      ck[k].c0       = c0;
      ck[k].g0       = in0 ? g[c0] : 0.0;
      ck[k].g1       = in1 ? g[c0 + 1] : 0.0;
      ck[k].x0 = ck[k].x1 = ck[k].h0 = ck[k].h1 = ck[k].u0 = ck[k].u1 = 0.0;
      if (pending) {
        if (in0) {
          ck[k].x0 = xi[c0];
          ck[k].h0 = hdg[c0];
          ck[k].u0 = uu[c0];
        }
        if (in1) {
          ck[k].x1 = xi[c0 + 1];
          ck[k].h1 = hdg[c0 + 1];
          ck[k].u1 = uu[c0 + 1];
        }
      }
      col[k][0] = col[k][1] = 0.0;
    }
    if (cBase < Rl) hess_range<NCH, false>(Hl, 0, cBase, Rl, wave, lane, ck, pending, rfac, fad, fae, xi, hdg, uu, g, rowsum, col);
    if (Rl < n) hess_range<NCH, PREFETCH>(Hg, Rl, max(Rl, cBase), n, wave, lane, ck, pending, rfac, fad, fae, xi, hdg, uu, g, rowsum, col);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {  // this wave's mirrored-entry sums of its columns: single writer
      if (ck[k].c0 < n) colsum[ck[k].c0] = col[k][0];
      if (ck[k].c0 + 1 < n) colsum[ck[k].c0 + 1] = col[k][1];
    }
  }
}

// Rows of up to 128 columns take one chunk per lane, longer rows two at a time (the benchmark sizes: n = 3 x 48 .. 4 x 96).
// PREFETCH: the HBM rows' next group is requested one group ahead (32 more VGPRs; off in the three-workgroups-per-CU kernels).
template <bool PREFETCH = true>
__device__ __forceinline__ void hess_pass(double* __restrict__ Hl, double* __restrict__ Hg, const int Rl, const int n, const bool pending,
                                          const double rfac, const double fad, const double fae, const double* __restrict__ xi,
                                          const double* __restrict__ hdg, const double* __restrict__ uu,
                                          const double* __restrict__ g, double* __restrict__ part) {
  if (n <= 128) {
    hess_pass_chunks<1, PREFETCH>(Hl, Hg, Rl, n, pending, rfac, fad, fae, xi, hdg, uu, g, part);
  } else {
    hess_pass_chunks<2, PREFETCH>(Hl, Hg, Rl, n, pending, rfac, fad, fae, xi, hdg, uu, g, part);
  }
}

// t = H g from the partial sums of hess_rows (fixed summation order).
__device__ __forceinline__ void hess_finish(const int n, const double* part, double* t) {
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += NT) {
    double v = part[i];
#pragma unroll
    for (int k = 1; k <= NW; ++k) v += part[k * n + i];
    t[i] = v;
  }
}

}  // namespace minim
}  // namespace nvmk
