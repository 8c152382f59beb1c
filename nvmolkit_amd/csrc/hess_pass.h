// The inverse-Hessian pass of the fused BFGS kernels (one workgroup per system) — shared by minimize.hip (through
// bfgs_device.inc) and tools/ubench_hess.hip (the pass is the hot spot of the conformer path; the microbenchmark runs it
// alone).  NO include guard: the file is compiled once per workgroup size, into the namespace NVMK_BFGS_NS (default: t256,
// 256 threads = four waves per system; minimize.hip also builds t64, one wave per system).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#ifndef NVMK_BFGS_NS
#define NVMK_BFGS_NS t256
#define NVMK_BFGS_THREADS 256
#define NVMK_BFGS_NS_DEFAULTED 1
#endif

namespace nvmk {
namespace minim {
namespace NVMK_BFGS_NS {

constexpr int NT = NVMK_BFGS_THREADS;  // threads of a BFGS workgroup (one workgroup per system)

// ---- inverse-Hessian pass -------------------------------------------------------------------------
// The inverse Hessian is symmetric.  Its DIAGONAL lives in an n-vector (in LDS for the whole minimisation); its strictly
// lower triangle is stored by rows, row r holding columns 0..r-1 padded to an even length (rows stay 16-byte aligned, the
// pad entry stays 0).  ONE pass per BFGS iteration streams the triangle once with the whole workgroup: it applies the
// rank-2 update that the PREVIOUS iteration left pending
//     H += rfac xi xi^T - fad hdg hdg^T + fae u u^T          (RDKit BFGSOpt.h; u = rfac xi - fad hdg)
// writes the element back, and accumulates t = H g for the current gradient.  The two products the textbook loop
// needs per iteration follow from t without touching the matrix again:
//     H dGrad       = H g_new - H g_old = t - hg                       (hg = H g of the previous iterate, kept in LDS)
//     H_new g_new   = t + rfac xi (xi.g) - fad hdg (hdg.g) + fae u (u.g)
// so the traffic per iteration is one read + one write of ~n^2 / 2 doubles instead of two reads + one write of n^2.
//
// Work split: ONE WAVE PER ROW.  Wave w owns rows w, w + NW, ...; its 64 lanes span 128 consecutive columns of the row as
// 16-byte pairs, so every load / store of a row is one fully coalesced (global) or conflict-free (LDS) wave instruction.
// A lane keeps the (pre-scaled) vector entries of ITS columns in registers; a row's coefficients are wave-uniform.  Four
// rows of a wave form a group: their loads are requested together, their four row sums are reduced together (DPP only),
// mirrored (column) sums stay in the lane's registers over all rows and are written once per wave.  Every partial sum has
// a single writer and every sum is formed in a fixed order that does not depend on where the rows live (LDS or HBM): a
// minimisation is reproducible bit for bit, whatever the launch's LDS budget.
//
// What bounds it (measured with tools/ubench_hess.hip, profiles/r02_conformers/): at these sizes (n = 36 .. 384) the pass is
// bound by INSTRUCTION ISSUE when the rows are in LDS and by the CU's share of HBM bandwidth when they are not — so the code
// below is written to minimise instructions per row: loads are unconditional (lanes past the end of a row read the next
// row / zeros and are masked out by ONE predicated region per row), the diagonal is kept apart so that every stored element
// is a mirrored one (no per-element special cases), rows shorter than 128 columns never touch the second column chunk.
// (Round 1 split rows over 8 row groups x 32 lanes with per-group partial sums in LDS: 20-27 us per pass at n = 144-192
// whether the matrix came from HBM or from LDS.)
constexpr int NW = NT / 64;  // waves per workgroup

__host__ __device__ __forceinline__ int64_t hess_row_offset(const int64_t r) {  // rows 0..r-1 (row k: k entries, padded to even)
  return (r * r) >> 1;
}
// the same inside one system (n < 46341): 32-bit scalar arithmetic in the pass's inner loop
__host__ __device__ __forceinline__ int hess_row_offset32(const int r) { return (r * r) >> 1; }

// Rows [0, Rl) of the packed triangle live in LDS behind the vectors (as many as the launch's LDS budget holds), rows Rl..
// stream from HBM.
__host__ __device__ __forceinline__ int resident_rows(const int n, const int64_t hldsDoubles) {
  if (hess_row_offset(n) <= hldsDoubles) return n;
  int r = 0;
  while (r < n && hess_row_offset(r + 1) <= hldsDoubles) ++r;
  // one-wave workgroups walk rows 0..63 in groups of eight that are all LDS or all HBM (hess_packed) and start every later
  // group of four on an even row (hess_range)
  if (NW == 1) r &= r < 64 ? ~7 : ~1;
  return r;
}
// LDS layout of bfgs_kernel: 10 vectors (the 10th is the diagonal of the inverse Hessian) + (1 + NW) partial-sum slabs of n
// doubles (row sums, then one slab of mirrored-entry sums per wave; the per-wave gradient slabs alias them), kRedDoubles of
// reduction scratch, then the resident rows of the inverse Hessian in whatever the launch's dynamic LDS (ldsDoubles) leaves.
constexpr int kRedDoubles = 8 * NW + 8;  // two alternating buffers of up to 4 values x NW waves (block reductions), padded
__host__ __device__ constexpr int64_t lds_vector_doubles(const int64_t n) { return (11 + NW) * n + kRedDoubles; }
__host__ __device__ constexpr int64_t lds_hessian_doubles(const int64_t ldsDoubles, const int64_t n) {
  return ldsDoubles > lds_vector_doubles(n) ? ldsDoubles - lds_vector_doubles(n) : 0;
}
// HBM bytes to allocate behind the last system: lanes past the end of the last rows read (and ignore) up to this much
constexpr int64_t kHessTailPadDoubles = 512;

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
  // the two cross-row steps on the VALU (gfx950 v_permlane16_swap / v_permlane32_swap of a register with its own copy) — an
  // LDS-crossbar shuffle here was two dependent ~100-cycle round trips per group of four rows
  {
    const auto l = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(x), false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(x), false, false);
    x            = __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);  // x + (lane ^ 16)
  }
  {
    const auto l = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(x), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(x), false, false);
    x            = __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);  // x + (lane ^ 32)
  }
  return x;
}

// State of one column chunk of 128 columns: this lane's two columns, the gradient entries and the PRE-SCALED update
// vectors that belong to them (xs = rfac xi, hs = fad hdg, us = fae u: a row then needs its raw xi, hdg, u only).
struct HessChunk {
  int    c0;
  double g0, g1, xs0, xs1, hs0, hs1, us0, us1;
};

// rowsum[i] += v by the row's single writer.  Sums in LDS (every class but the HBM-vector one): one ds_add_f64 — no read, no
// wait for it — which rounds exactly like the read / add / write it replaces.
template <bool LDS_SUMS> __device__ __forceinline__ void add_to_sum(double* __restrict__ sum, const double v) {
  if constexpr (LDS_SUMS) {
    __hip_atomic_fetch_add(sum, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else {
    *sum += v;
  }
}

// Rows [rFrom, rEnd) of NCH adjacent column chunks; the matrix row `rBase` starts at H (LDS or HBM: the address space is
// known at the call).  With PREFETCH (HBM rows) the NEXT group's matrix pairs are requested before the current group is
// worked on — ahead of the current group's stores, so waiting for them does not wait for the stores (vmcnt is in-order).
// The wave index is scalar: row numbers, row offsets and the branches on them live on the scalar unit.
// col[k][0..1] (mirrored-entry sums of the lane's columns) are carried by the caller across ranges.
// Groups of four rows that lie entirely inside the range take a copy of the group code without the per-row range tests; the
// last, partial group takes the one with them.
// AHEAD > 0 (the team class, whose rows all come from HBM through one CU's eight waves): beyond the group requested into
// registers, the groups AHEAD further on are TOUCHED — one 4-byte load per 64-byte sector, L1-bypassing, its result never used —
// so that they are on their way from HBM to the L2 while the registers can hold no more: bytes in flight per wave without a
// register per 16 of them.
#ifndef NVMK_HESS_AHEAD
#define NVMK_HESS_AHEAD 0
#endif
template <int NCH, bool PREFETCH, bool LDS_SUMS, int AHEAD = 0>
__device__ __forceinline__ void hess_range(double* __restrict__ H, const int rBase, const int rFrom, const int rEnd, const int wave,
                                           const int lane, const HessChunk (&ck)[NCH], const bool pending,
                                           const double* __restrict__ xi, const double* __restrict__ hdg,
                                           const double* __restrict__ uu, const double* __restrict__ g, double* __restrict__ rowsum,
                                           double (&col)[NCH][2]) {
  constexpr int RU   = 4;
  const int     base = hess_row_offset32(rBase);
  auto entry = [&](const int r, const int c0) -> double* { return H + (hess_row_offset32(r) - base) + c0; };
  auto load_group = [&](const int r0, double2 (&dst)[RU][NCH], auto fullTag) {
    constexpr bool FULL = decltype(fullTag)::value;
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int r = FULL ? r0 + NW * u : min(r0 + NW * u, rEnd - 1);  // rows past the range re-read the last one (never used)
#pragma unroll
      for (int k = 0; k < NCH; ++k) dst[u][k] = *reinterpret_cast<const double2*>(entry(r, ck[k].c0));  // unconditional
    }
  };
  auto work = [&](const int r0, const double2 (&hv)[RU][NCH], auto fullTag) {
    constexpr bool FULL = decltype(fullTag)::value;
    // the rows' coefficients: ONE address per vector and constant offsets (paired LDS reads).  Rows past rEnd read whatever
    // follows in the vector block (every vector is followed by another one) and are never used.
    double        gr[RU], xr[RU], hr[RU], ur[RU], rs[RU];
    const double *gp = g + r0, *xp = xi + r0, *hp = hdg + r0, *up = uu + r0;
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      gr[u] = gp[NW * u];
      xr[u] = xp[NW * u];
      hr[u] = hp[NW * u];
      ur[u] = up[NW * u];
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int r = r0 + NW * u;
      rs[u]       = 0.0;
      if (FULL || r < rEnd) {  // wave-uniform
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          const int c0 = ck[k].c0;
          if (c0 < r) {  // ONE predicated region per row and chunk: the lanes that hold entries of this row
            double2 h = hv[u][k];
            if (pending) {
              h.x += xr[u] * ck[k].xs0 - hr[u] * ck[k].hs0 + ur[u] * ck[k].us0;
              const double y = h.y + (xr[u] * ck[k].xs1 - hr[u] * ck[k].hs1 + ur[u] * ck[k].us1);
              // the pad entry of an odd-length row stays 0 (its lane: c0 + 1 == r).  One-wave workgroups start their groups
              // on even rows (hess_pass), so every other row of a group has an even length and no pad at all.
              h.y = (NW == 1 && (u & 1) == 0) ? y : (c0 + 1 < r ? y : 0.0);
              *reinterpret_cast<double2*>(entry(r, c0)) = h;
            }
            col[k][0] += h.x * gr[u];  // every stored entry is strictly below the diagonal: it has a mirror image
            col[k][1] += h.y * gr[u];
            rs[u] += h.x * ck[k].g0 + h.y * ck[k].g1;
          }
        }
      }
    }
    // the row this LANE adds the sum of (lane & 3 selects it, see wave_sum4_transposed): one writer per row (this wave),
    // column super-chunks in order
    const double tot   = wave_sum4_transposed(rs, lane);
    const int    myRow = r0 + NW * (lane & 3);
    if (lane < 4 && (FULL || myRow < rEnd)) add_to_sum<LDS_SUMS>(rowsum + myRow, tot);
  };
  int           r0   = rFrom + ((wave - rFrom) % NW + NW) % NW;  // first row of this wave at or after rFrom
  constexpr int STEP = NW * RU;
  // the sectors of one group: RU rows x NCH KiB = RU * NCH * 16 sectors of 64 bytes, 64 per wave instruction
  [[maybe_unused]] auto touch_group = [&](const int g0) {
    if constexpr (AHEAD > 0) {
      constexpr int kInstr = RU * NCH * 16 / 64;  // RU = 4: NCH instructions
#pragma unroll
      for (int j = 0; j < kInstr; ++j) {
        const int s   = j * 64 + lane;            // sector of the group
        const int u   = s / (NCH * 16), q = s % (NCH * 16);
        const int r   = min(g0 + NW * u, rEnd - 1);
        const double* p = H + (hess_row_offset32(r) - base) + ck[0].c0 - 2 * lane + 8 * q;
        (void)__hip_atomic_load(reinterpret_cast<const unsigned int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  if constexpr (PREFETCH) {
    double2 next[RU][NCH];
    if (r0 < rEnd) load_group(r0, next, std::false_type{});
    if constexpr (AHEAD > 0) {
#pragma unroll
      for (int a = 1; a < AHEAD; ++a)
        if (r0 + a * STEP < rEnd) touch_group(r0 + a * STEP);
    }
    for (; r0 + NW * (RU - 1) < rEnd; r0 += STEP) {
      double2 hv[RU][NCH];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) hv[u][k] = next[u][k];
      }
      if (r0 + STEP < rEnd) load_group(r0 + STEP, next, std::false_type{});
      if constexpr (AHEAD > 0) {
        if (r0 + (1 + AHEAD) * STEP - STEP < rEnd) touch_group(r0 + AHEAD * STEP);
      }
      work(r0, hv, std::true_type{});
    }
    if (r0 < rEnd) work(r0, next, std::false_type{});
  } else {
    for (; r0 + NW * (RU - 1) < rEnd; r0 += STEP) {
      double2 hv[RU][NCH];
      load_group(r0, hv, std::true_type{});
      work(r0, hv, std::true_type{});
    }
    if (r0 < rEnd) {
      double2 hv[RU][NCH];
      load_group(r0, hv, std::false_type{});
      work(r0, hv, std::false_type{});
    }
  }
}

// One-wave workgroups only: rows [rFrom, rEnd) inside 0..63 (both multiples of 8), TWO ROWS PER WAVE INSTRUCTION.  A row
// shorter than 64 columns uses at most half a wave in hess_range; here lanes 0..31 take row r and lanes 32..63 row r + 1,
// lane & 31 = the same pair of columns in both halves, four such row pairs (eight rows) per group.  Row sums are reduced
// inside each half (wave_sum4_transposed without its last step), the mirrored-entry sums of the two halves are folded into
// colOut[0..1] of lanes 0..31 at the end.  Halves the instructions these rows cost (a third of the rows at n = 144).
template <bool PREFETCH, bool LDS_SUMS>
__device__ __forceinline__ void hess_packed(double* __restrict__ H, const int rBase, const int rFrom, const int rEnd, const int lane,
                                            const HessChunk& ck, const bool pending, const double* __restrict__ xi,
                                            const double* __restrict__ hdg, const double* __restrict__ uu,
                                            const double* __restrict__ g, double* __restrict__ rowsum, double (&colAcc)[2]) {
  constexpr int RU   = 2;  // row pairs per group (four rows): two keep the one-wave kernels free of spills
  const int     half = lane >> 5;
  const int     base = hess_row_offset32(rBase);
  const int     c0   = ck.c0;  // 2 * (lane & 31)
  auto load_group = [&](const int r0, double2 (&dst)[RU]) {
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int r = r0 + 2 * u + half;
      dst[u]      = *reinterpret_cast<const double2*>(H + (hess_row_offset32(r) - base) + c0);  // unconditional (tail pad / next rows)
    }
  };
  double2 next[RU];
  if constexpr (PREFETCH) {
    if (rFrom < rEnd) load_group(rFrom, next);
  }
  for (int r0 = rFrom; r0 < rEnd; r0 += 2 * RU) {
    double2 hv[RU];
    if constexpr (PREFETCH) {
#pragma unroll
      for (int u = 0; u < RU; ++u) hv[u] = next[u];
      if (r0 + 2 * RU < rEnd) load_group(r0 + 2 * RU, next);
    } else {
      load_group(r0, hv);
    }
    double rs[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int    r  = r0 + 2 * u + half;
      const double gr = g[r], xr = xi[r], hr = hdg[r], ur = uu[r];
      rs[u]           = 0.0;
      if (c0 < r) {
        double2    h   = hv[u];
        const bool two = c0 + 1 < r;
        if (pending) {
          h.x += xr * ck.xs0 - hr * ck.hs0 + ur * ck.us0;
          const double y = h.y + (xr * ck.xs1 - hr * ck.hs1 + ur * ck.us1);
          h.y            = two ? y : 0.0;
          *reinterpret_cast<double2*>(H + (hess_row_offset32(r) - base) + c0) = h;
        }
        colAcc[0] += h.x * gr;
        colAcc[1] += h.y * gr;
        rs[u] = h.x * ck.g0 + h.y * ck.g1;
      }
    }
    // two row sums per half, transposed: even lanes end with row pair 0's total of their half, odd lanes with row pair 1's
    {
      const bool odd = (lane & 1) != 0;
      double     x   = (odd ? rs[1] : rs[0]) + dpp_mov<0xb1>(odd ? rs[0] : rs[1]);  // lane & 1 = u: sum over the lane pair
      x += dpp_mov<0x4e>(x);   // quad
      x += dpp_mov<0x124>(x);  // row_ror 4
      x += dpp_mov<0x128>(x);  // row_ror 8: the row of 16 lanes
      {
        const auto l = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(x), false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(x), false, false);
        x            = __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);  // x + (lane ^ 16): the half's total
      }
      const int myRow = r0 + 2 * (lane & 1) + half;
      if ((lane & 31) < 2) add_to_sum<LDS_SUMS>(rowsum + myRow, x);  // one writer per row: this wave, once
    }
  }
}

template <int NCH> __device__ __forceinline__ void hess_chunk_state(HessChunk (&ck)[NCH], double (&col)[NCH][2], const int cBase,
                                                                    const int lane, const int n, const bool pending, const double rfac,
                                                                    const double fad, const double fae, const double* __restrict__ xi,
                                                                    const double* __restrict__ hdg, const double* __restrict__ uu,
                                                                    const double* __restrict__ g) {
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int  c0  = cBase + 128 * k + 2 * lane;
    const bool in0 = c0 < n, in1 = c0 + 1 < n;
    ck[k].c0       = c0;
    ck[k].g0       = in0 ? g[c0] : 0.0;
    ck[k].g1       = in1 ? g[c0 + 1] : 0.0;
    ck[k].xs0 = ck[k].xs1 = ck[k].hs0 = ck[k].hs1 = ck[k].us0 = ck[k].us1 = 0.0;
    if (pending) {
      if (in0) {
        ck[k].xs0 = rfac * xi[c0];
        ck[k].hs0 = fad * hdg[c0];
        ck[k].us0 = fae * uu[c0];
      }
      if (in1) {
        ck[k].xs1 = rfac * xi[c0 + 1];
        ck[k].hs1 = fad * hdg[c0 + 1];
        ck[k].us1 = fae * uu[c0 + 1];
      }
    }
    col[k][0] = col[k][1] = 0.0;
  }
}

// The pass: diagonal in `diag` (LDS), rows [0, Rl) of the strict lower triangle from LDS (Hl), rows [Rl, n) from HBM (Hg,
// whose first element is row Rl's).  `part` = row sums [n] then NW slabs [n] of mirrored-entry sums (all written here).
// PREFETCH: the HBM rows' next group is requested one group ahead (more VGPRs; the microbenchmark can switch it off).
// Must be entered by the whole workgroup after a barrier (it starts by writing the row sums of the diagonal).
template <bool PREFETCH = true, bool LDS_SUMS = true>
__device__ __forceinline__ void hess_pass(double* __restrict__ diag, double* __restrict__ Hl, double* __restrict__ Hg, const int Rl,
                                          const int n, const bool pending, const double rfac, const double fad, const double fae,
                                          const double* __restrict__ xi, const double* __restrict__ hdg, const double* __restrict__ uu,
                                          const double* __restrict__ g, double* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  double*   rowsum = part;
  double*   colsum = part + (1 + wave) * n;
  // the diagonal: update, and start every row sum with its term
  for (int i = threadIdx.x; i < n; i += NT) {
    double d = diag[i];
    if (pending) {
      d += rfac * xi[i] * xi[i] - fad * hdg[i] * hdg[i] + fae * uu[i] * uu[i];
      diag[i] = d;
    }
    rowsum[i] = d * g[i];
  }
  if constexpr (NW == 1) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  } else {
    __syncthreads();
  }
  // one-wave workgroups: rows 0 .. P-1 (P <= 64, a multiple of 8) two at a time; their mirrored-entry sums start the first
  // chunk's column accumulators of lanes 0..31 below
  int    packedRows  = 0;
  double packedCol[2] = {0.0, 0.0};
  if constexpr (NW == 1) {
    packedRows = min(n, 64) & ~7;  // (whole groups of four rows, and a boundary resident_rows can land on)
    if (packedRows > 0) {
      HessChunk ckp[1];
      double    colp[1][2];
      // the chunk state of columns 2 (lane & 31), 2 (lane & 31) + 1 — the first 64 columns, in both halves of the wave
      hess_chunk_state<1>(ckp, colp, 0, lane & 31, n, pending, rfac, fad, fae, xi, hdg, uu, g);
      const int split = min(packedRows, Rl);  // a multiple of 8 (resident_rows)
      if (split > 0) hess_packed<false, LDS_SUMS>(Hl, 0, 0, split, lane, ckp[0], pending, xi, hdg, uu, g, rowsum, colp[0]);
      if (split < packedRows) hess_packed<PREFETCH, LDS_SUMS>(Hg, Rl, split, packedRows, lane, ckp[0], pending, xi, hdg, uu, g, rowsum, colp[0]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {  // fold the two halves: lanes 0..31 keep the totals of their columns
        const auto l = __builtin_amdgcn_permlane32_swap(__double2loint(colp[0][j]), __double2loint(colp[0][j]), false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(__double2hiint(colp[0][j]), __double2hiint(colp[0][j]), false, false);
        const double t = __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
        packedCol[j]   = lane < 32 ? t : 0.0;
      }
    }
  }
  for (int cBase = 0; cBase < n; cBase += 256) {  // column super-chunk of 2 x 128 columns: rows before cBase have no column in it
    HessChunk ck[2];
    double    col[2][2];
    hess_chunk_state<2>(ck, col, cBase, lane, n, pending, rfac, fad, fae, xi, hdg, uu, g);
    if (cBase == 0) {
      col[0][0] = packedCol[0];
      col[0][1] = packedCol[1];
    }
    // rows that reach into the first chunk only ([cBase, cBase + 128]) never touch the second one
    // row cBase + 129 is the first with an entry in the second chunk (c0 < r); one-wave workgroups split one row earlier so
    // that both ranges start on an even row (row cBase + 128 then finds no lane of the second chunk: correct, one idle step)
    const int mid = min(n, cBase + (NW == 1 ? 128 : 129));
    {
      HessChunk(&ck1)[1]    = reinterpret_cast<HessChunk(&)[1]>(ck[0]);
      double(&col1)[1][2]   = reinterpret_cast<double(&)[1][2]>(col[0]);
      const int lo = max(cBase, packedRows), hi = mid;
      if (lo < min(hi, Rl)) hess_range<1, false, LDS_SUMS>(Hl, 0, lo, min(hi, Rl), wave, lane, ck1, pending, xi, hdg, uu, g, rowsum, col1);
      if (max(lo, Rl) < hi) hess_range<1, PREFETCH, LDS_SUMS>(Hg, Rl, max(lo, Rl), hi, wave, lane, ck1, pending, xi, hdg, uu, g, rowsum, col1);
    }
    if (mid < n) {
      if (mid < Rl) hess_range<2, false, LDS_SUMS>(Hl, 0, mid, Rl, wave, lane, ck, pending, xi, hdg, uu, g, rowsum, col);
      if (max(mid, Rl) < n) hess_range<2, PREFETCH, LDS_SUMS>(Hg, Rl, max(mid, Rl), n, wave, lane, ck, pending, xi, hdg, uu, g, rowsum, col);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // this wave's mirrored-entry sums of its columns: single writer
      if (ck[k].c0 < n) colsum[ck[k].c0] = col[k][0];
      if (ck[k].c0 + 1 < n) colsum[ck[k].c0 + 1] = col[k][1];
    }
  }
}

// ---- the pass of ONE RANK of a cooperative team (bfgs_team_kernel: several workgroups per system) ---------------------------------
// The rows of the triangle are dealt over the team's workgroups in contiguous blocks of about equal cost; this workgroup walks
// rows [Ra, Rb) of the whole triangle in HBM (Hg, row 0 first) and leaves what it contributes to t = H g in LDS:
//   colTot[i], i < Rb   the mirrored-entry sums of column i over its rows (the waves' sums added in wave order),
//   rowSum(r), Ra <= r < Rb  the diagonal's term + row r's sum          (team_pass_contribution below reads both).
// The vectors themselves (diag, xi, hdg, uu, g) are the workgroup's own full copies in HBM; what a ROW needs of them (its four
// coefficients) and the row sums live in LDS for the duration of the pass, because in HBM every group of four rows waits for its
// coefficient loads and for its sum's read-modify-write, and the hardware returns loads in order, so that wait also drains the
// matrix rows requested ahead (the HBM-vector class without a team streams at 12 GB/s per workgroup for this reason).  The waves'
// mirrored-entry sums of a 256-column chunk meet in LDS too (two scratch blocks in turn, one barrier per chunk) instead of in
// per-wave n-vectors in HBM.  LDS: colTot = n doubles; stage = 5 x (rowsCap + kTeamStagePad) + 2 x NW x 256 doubles.  Blocks
// taller than rowsCap are walked in several sub-blocks (the row sums of a sub-block are then moved to `rowOut`, an n-vector in
// HBM, instead of staying in LDS: `rowsInLds` tells the caller).  The diagonal is updated by every rank on its own copy.
constexpr int kTeamStagePad = 32;  // hess_range reads the coefficients of up to 3 NW rows past a range's end (and never uses them)
__host__ __device__ constexpr int team_pass_stage_doubles(const int rowsCap) { return 5 * (rowsCap + kTeamStagePad) + 2 * NW * 256; }
template <bool PREFETCH = true>
__device__ __forceinline__ void hess_pass_rows(double* __restrict__ diag, double* __restrict__ Hg, const int Ra, const int Rb, const int n,
                                               const bool pending, const double rfac, const double fad, const double fae,
                                               const double* __restrict__ xi, const double* __restrict__ hdg,
                                               const double* __restrict__ uu, const double* __restrict__ g, double* __restrict__ colTot,
                                               double* __restrict__ stage, const int rowsCap, double* __restrict__ rowOut) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  if (pending) {  // (batches of four elements: the loads of a batch are in flight together)
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * NT) {
      double d[4], x[4], h[4], u[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = min(i0 + k * NT, n - 1);
        d[k] = diag[i], x[k] = xi[i], h[k] = hdg[i], u[k] = uu[i];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (i0 + k * NT < n) diag[i0 + k * NT] = d[k] + (rfac * x[k] * x[k] - fad * h[k] * h[k] + fae * u[k] * u[k]);
      }
    }
  }
  const int stride = rowsCap + kTeamStagePad;
  double *sg = stage, *sx = stage + stride, *sh = stage + 2 * stride, *su = stage + 3 * stride, *ss = stage + 4 * stride;
  double*   scratch = stage + 5 * stride;  // 2 x NW x 256
  int       flip    = 0;
  for (int ra = Ra; ra < Rb; ra += rowsCap) {
    const int rb = min(Rb, ra + rowsCap);
    __syncthreads();  // the previous sub-block's sums have left the staging area (first time round: whoever used this LDS before is done, and every diag entry is written)
    for (int i0 = threadIdx.x; i0 < rb - ra + kTeamStagePad; i0 += 2 * NT) {
      double v[2][5];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int r = min(ra + i0 + k * NT, rb - 1);
        v[k][0] = g[r], v[k][1] = xi[r], v[k][2] = hdg[r], v[k][3] = uu[r], v[k][4] = diag[r];
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = i0 + k * NT;
        if (i < rb - ra + kTeamStagePad) {
          const bool in = ra + i < rb;
          sg[i] = in ? v[k][0] : 0.0;
          sx[i] = in ? v[k][1] : 0.0;
          sh[i] = in ? v[k][2] : 0.0;
          su[i] = in ? v[k][3] : 0.0;
          ss[i] = in ? v[k][4] * v[k][0] : 0.0;
        }
      }
    }
    __syncthreads();
    // (the first sub-block also visits the column chunks only later sub-blocks have rows in, so that every column below Rb starts
    // from zero in this pass)
    for (int cBase = 0; cBase < (ra == Ra ? Rb : rb); cBase += 256) {
      HessChunk ck[2];
      double    col[2][2];
      hess_chunk_state<2>(ck, col, cBase, lane, n, pending, rfac, fad, fae, xi, hdg, uu, g);
      const int mid = min(rb, cBase + 129);  // rows before it have no entry in the second chunk
      {
        HessChunk(&ck1)[1]  = reinterpret_cast<HessChunk(&)[1]>(ck[0]);
        double(&col1)[1][2] = reinterpret_cast<double(&)[1][2]>(col[0]);
        const int lo = max(cBase, ra);
        if (lo < mid) hess_range<1, PREFETCH, true, NVMK_HESS_AHEAD>(Hg, 0, lo, mid, wave, lane, ck1, pending, sx - ra, sh - ra, su - ra, sg - ra, ss - ra, col1);
      }
      const int lo2 = max(mid, ra);
      if (lo2 < rb) hess_range<2, PREFETCH, true, NVMK_HESS_AHEAD>(Hg, 0, lo2, rb, wave, lane, ck, pending, sx - ra, sh - ra, su - ra, sg - ra, ss - ra, col);
      // the waves' sums of this chunk's 256 columns, added in wave order by the column's thread
      double* sc = scratch + flip * (NW * 256);
      flip ^= 1;
#pragma unroll
      for (int k = 0; k < 2; ++k) *reinterpret_cast<double2*>(sc + wave * 256 + 128 * k + 2 * lane) = make_double2(col[k][0], col[k][1]);
      __syncthreads();  // (one per chunk: the block written two chunks ago was read before the previous chunk's barrier)
      if (threadIdx.x < 256 && cBase + static_cast<int>(threadIdx.x) < n) {
        double v = sc[threadIdx.x];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += sc[w * 256 + threadIdx.x];
        const int c = cBase + static_cast<int>(threadIdx.x);
        colTot[c]   = ra == Ra ? v : colTot[c] + v;
      }
    }
    __syncthreads();
    if (Rb - Ra > rowsCap) {  // several sub-blocks: the row sums move out before the staging area is refilled
      for (int i = threadIdx.x; i < rb - ra; i += NT) rowOut[ra + i] = ss[i];
    }
  }
}
// The same pass with the COLUMN vectors in LDS (round 6, second form): where n + 4 (Rb + pad) + the block's rows + the scratch fit
// the rank's LDS, the four vectors the pass reads — g, xi, H dg, u for columns [0, Rb) — are staged once per pass (in the loop that
// updates the diagonal anyway) and every wave takes the state of a 256-column chunk from there; the rows' coefficients are entries
// of the same arrays.  Without it each of the eight waves reads all four vectors from HBM / L2 once per pass: 8 x 4 x 8 n bytes per
// rank, ~12 % of what the rank streams of the triangle itself, replicated on every rank of the team.
// LDS: colTot = n doubles; stage = 4 x (Rb + kTeamStagePad) + (Rb - Ra + kTeamStagePad) + 2 x NW x 256 doubles; one sub-block.
__host__ __device__ constexpr int team_pass_cols_doubles(const int Ra, const int Rb) {
  return 4 * (Rb + kTeamStagePad) + (Rb - Ra + kTeamStagePad) + 2 * NW * 256;
}
template <bool PREFETCH = true>
__device__ __forceinline__ void hess_pass_rows_cols(double* __restrict__ diag, double* __restrict__ Hg, const int Ra, const int Rb, const int n,
                                                    const bool pending, const double rfac, const double fad, const double fae,
                                                    const double* __restrict__ xi, const double* __restrict__ hdg,
                                                    const double* __restrict__ uu, const double* __restrict__ g,
                                                    double* __restrict__ colTot, double* __restrict__ stage) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int cs   = Rb + kTeamStagePad;
  double *cg = stage, *cx = stage + cs, *ch = stage + 2 * cs, *cu = stage + 3 * cs, *ss = stage + 4 * cs;
  double* scratch = ss + (Rb - Ra + kTeamStagePad);
  __syncthreads();  // whoever used this LDS before is done
  // one walk over the vectors: the diagonal's update (every rank, all n), the column vectors and the row sums' first terms into LDS
  for (int i0 = threadIdx.x; i0 < max(n, cs); i0 += 4 * NT) {
    double d[4], x[4], h[4], u[4], gg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = min(i0 + k * NT, n - 1);
      d[k] = diag[i], x[k] = xi[i], h[k] = hdg[i], u[k] = uu[i], gg[k] = g[i];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k * NT;
      if (i < n) {
        double dd = d[k];
        if (pending) {
          dd += rfac * x[k] * x[k] - fad * h[k] * h[k] + fae * u[k] * u[k];
          diag[i] = dd;
        }
        if (i >= Ra && i < Rb) ss[i - Ra] = dd * gg[k];
      }
      if (i < cs) {
        const bool in = i < Rb;
        cg[i] = in ? gg[k] : 0.0;
        cx[i] = in ? x[k] : 0.0;
        ch[i] = in ? h[k] : 0.0;
        cu[i] = in ? u[k] : 0.0;
      }
    }
  }
  if (threadIdx.x < kTeamStagePad) ss[Rb - Ra + threadIdx.x] = 0.0;
  __syncthreads();
  int flip = 0;
  for (int cBase = 0; cBase < Rb; cBase += 256) {
    HessChunk ck[2];
    double    col[2][2];
    hess_chunk_state<2>(ck, col, cBase, lane, Rb, pending, rfac, fad, fae, cx, ch, cu, cg);  // (columns at or beyond Rb have no entry in these rows)
    const int mid = min(Rb, cBase + 129);
    {
      HessChunk(&ck1)[1]  = reinterpret_cast<HessChunk(&)[1]>(ck[0]);
      double(&col1)[1][2] = reinterpret_cast<double(&)[1][2]>(col[0]);
      const int lo = max(cBase, Ra);
      if (lo < mid) hess_range<1, PREFETCH, true, NVMK_HESS_AHEAD>(Hg, 0, lo, mid, wave, lane, ck1, pending, cx, ch, cu, cg, ss - Ra, col1);
    }
    const int lo2 = max(mid, Ra);
    if (lo2 < Rb) hess_range<2, PREFETCH, true, NVMK_HESS_AHEAD>(Hg, 0, lo2, Rb, wave, lane, ck, pending, cx, ch, cu, cg, ss - Ra, col);
    double* sc = scratch + flip * (NW * 256);
    flip ^= 1;
#pragma unroll
    for (int k = 0; k < 2; ++k) *reinterpret_cast<double2*>(sc + wave * 256 + 128 * k + 2 * lane) = make_double2(col[k][0], col[k][1]);
    __syncthreads();
    if (threadIdx.x < 256 && cBase + static_cast<int>(threadIdx.x) < n) {
      double v = sc[threadIdx.x];
#pragma unroll
      for (int w = 1; w < NW; ++w) v += sc[w * 256 + threadIdx.x];
      colTot[cBase + static_cast<int>(threadIdx.x)] = v;
    }
  }
  __syncthreads();
}
__device__ __forceinline__ double team_pass_contribution_cols(const int i, const int Ra, const int Rb, const double* __restrict__ colTot,
                                                              const double* __restrict__ stage) {
  if (i >= Rb) return 0.0;
  double v = colTot[i];
  if (i >= Ra) v += stage[4 * (Rb + kTeamStagePad) + (i - Ra)];
  return v;
}

// What this rank adds to (H g)[i] after hess_pass_rows: the column's sums, and for its own rows the row's.
__device__ __forceinline__ double team_pass_contribution(const int i, const int Ra, const int Rb, const int rowsCap, const double* __restrict__ colTot,
                                                         const double* __restrict__ stage, const double* __restrict__ rowOut) {
  if (i >= Rb) return 0.0;
  double v = colTot[i];
  if (i >= Ra) v += (Rb - Ra > rowsCap) ? rowOut[i] : stage[4 * (rowsCap + kTeamStagePad) + (i - Ra)];
  return v;
}

// t = H g from the partial sums of the pass (fixed summation order).
__device__ __forceinline__ void hess_finish(const int n, const double* part, double* t) {
  if constexpr (NW == 1) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  } else {
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n; i += NT) {
    double v = part[i];
#pragma unroll
    for (int k = 1; k <= NW; ++k) v += part[k * n + i];
    t[i] = v;
  }
}

}  // namespace NVMK_BFGS_NS
}  // namespace minim
}  // namespace nvmk

#ifdef NVMK_BFGS_NS_DEFAULTED
#undef NVMK_BFGS_NS
#undef NVMK_BFGS_THREADS
#undef NVMK_BFGS_NS_DEFAULTED
#endif
