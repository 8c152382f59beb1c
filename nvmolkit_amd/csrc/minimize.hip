// Batched force-field evaluation and fused per-system BFGS minimisation — gfx950.
//
// Replaces (reference paths):
//   src/minimizer/bfgs_minimize_permol_kernels.cu:35-745   bfgsMinimizeKernel (whole BFGS in one launch)
//   src/minimizer/bfgs_minimize.cu:978-1084                BfgsBatchMinimizer::minimize (host-driven variant)
//   src/forcefields/{dist_geom,mmff}_kernels*.cu           combinedEnergies / combinedGrad block-per-molecule kernels
// The optimiser is RDKit's BFGS (the reference restates ForceFields/BFGSOpt.h): identity inverse Hessian,
// gradient scaling 0.1 then halving while max > 10, backtracking cubic line search (FUNCTOL 1e-4, MOVETOL 1e-7,
// at most 1000 steps), TOLX = 1.2e-7, BFGS update guarded by fac^2 > EPS |dg|^2 |xi|^2 with EPS = 3e-8
// (bfgs_minimize_permol_kernels.cu:29-33, :304-407).
//
// MI355X design: one 256-thread workgroup per system; positions, gradient, direction, trial positions and
// gradient difference live in LDS for the whole minimisation; all arithmetic is fp64; term tables are read
// straight from HBM/L2 (they are shared by the conformers of a molecule); the inverse Hessian is the only
// per-system state in HBM.  Term loops are thread-strided over a generic table layout (see nvmolkit_amd.h):
// every term group is {CSR starts, interleaved local atom indices, interleaved double parameters}.
#include <algorithm>
#include <vector>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "options.h"
#include "ff_grad.h"
#include "hess_pass.h"
#include "ff_terms.h"

namespace nvmk {
namespace minim {

using namespace nvmk::ff;

struct Group {
  const int32_t* starts;
  const int32_t* idx;
  const double*  par;
};
struct Batch {
  int            kind;
  int            nSystems;
  const int32_t* atomStarts;
  Group          g[12];
  const int32_t* sysMol;      // optional: term tables are per MOLECULE and system s uses row sysMol[s] of every `starts`
  unsigned       groupMask;   // bit g set = evaluate term group g
  // ETK only, optional: per-system reference distances for the 1-2 / 1-3 restraints (the reference re-centres
  // those bounds on the current geometry before the ETK minimisation, etkdg_stage_etk_minimization.cu:32-64)
  const int32_t* refStarts[2];
  const double*  ref[2];
};

// Internal kinds: MMFF / UFF batches that carry constraint groups run separate kernel instantiations, so the common
// unconstrained kernels keep their register budget (the reference templates its kernels on HasConstraints).
constexpr int KIND_MMFF_C = 5;
constexpr int KIND_UFF_C  = 6;
template <int KIND> struct Dim {
  static constexpr int value = (KIND == NVMK_FF_DG || KIND == NVMK_FF_QUARTIC) ? 4 : 3;
};

// ---- block reductions -----------------------------------------------------------------------------
enum class Op { kSum, kMax, kMin };
template <Op OP> __device__ __forceinline__ double combine(const double a, const double b) {
  if constexpr (OP == Op::kSum) return a + b;
  if constexpr (OP == Op::kMax) return a > b ? a : b;
  return a < b ? a : b;
}
// All threads receive the result.  `red` is NT/64 + 1 doubles of LDS.
template <Op OP> __device__ __forceinline__ double block_reduce(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = combine<OP>(v, __shfl_xor(v, o));
  __syncthreads();  // protect `red` from the previous use
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = red[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) r = combine<OP>(r, red[w]);
  return r;
}

// N sums at once: one pair of barriers instead of N (the kernel is barrier-bound between its short vector loops:
// SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = 0.19, profiles/r02_conformers_round1_pass).  `red` holds (NT / 64) * N doubles.
// The summation order of each value is the same as block_reduce<kSum>'s.
template <int N> __device__ __forceinline__ void block_sum_n(double (&v)[N], double* red) {
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
  }
  __syncthreads();  // protect `red` from the previous use
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) red[(threadIdx.x >> 6) * N + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double r = red[k];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) r += red[w * N + k];
    v[k] = r;
  }
}

// Block reductions of the fused BFGS kernel (about ten per iteration; the kernel is latency / barrier bound).  Two
// things make them cheaper than block_reduce above: the wave stage runs on the DPP crossbar (quad, row of 16, then the
// four rows through scalar registers: no LDS-crossbar permutes, which cost ~100 cycles each in a dependent chain of six)
// and the LDS stage alternates between two buffers, so ONE barrier per reduction suffices — a thread can only reach the
// write of reduction k + 2 after everyone has passed the barrier of reduction k + 1, i.e. after all reads of reduction k.
template <Op OP> __device__ __forceinline__ double wave_reduce_dpp(double v) {
  v = combine<OP>(v, dpp_mov<0xb1>(v));   // quad_perm [1, 0, 3, 2]
  v = combine<OP>(v, dpp_mov<0x4e>(v));   // quad_perm [2, 3, 0, 1]
  v = combine<OP>(v, dpp_mov<0x124>(v));  // row_ror 4
  v = combine<OP>(v, dpp_mov<0x128>(v));  // row_ror 8: every lane holds its row's result
  double r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * k), __builtin_amdgcn_readlane(__double2loint(v), 16 * k));
  }
  return combine<OP>(combine<OP>(r[0], r[1]), combine<OP>(r[2], r[3]));
}
struct BlockReducer {
  double* red;    // kRedDoubles of LDS
  int     phase;  // which half the next reduction uses (uniform)
  template <Op OP> __device__ __forceinline__ double run(double v) {
    v            = wave_reduce_dpp<OP>(v);
    double* slot = red + phase * (kRedDoubles / 2);
    phase ^= 1;
    if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = slot[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r = combine<OP>(r, slot[w]);
    return r;
  }
  // NS sums followed by NM maxima in one barrier: v[0 .. NS) are summed, v[NS .. NS + NM) maximised
  template <int NS, int NM> __device__ __forceinline__ void sums_and_maxima(double (&v)[NS + NM]) {
    constexpr int N = NS + NM;
    static_assert(N * NW <= kRedDoubles / 2, "reduction scratch too small");
    double* slot = red + phase * (kRedDoubles / 2);
    phase ^= 1;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      v[k] = k < NS ? wave_reduce_dpp<Op::kSum>(v[k]) : wave_reduce_dpp<Op::kMax>(v[k]);
      if ((threadIdx.x & 63) == 0) slot[(threadIdx.x >> 6) * N + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
      double r = slot[k];
#pragma unroll
      for (int w = 1; w < NW; ++w) r = k < NS ? r + slot[w * N + k] : fmax(r, slot[w * N + k]);
      v[k] = r;
    }
  }
  template <int N> __device__ __forceinline__ void sum_n(double (&v)[N]) {
    static_assert(N * NW <= kRedDoubles / 2, "reduction scratch too small");
    double* slot = red + phase * (kRedDoubles / 2);
    phase ^= 1;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      v[k] = wave_reduce_dpp<Op::kSum>(v[k]);
      if ((threadIdx.x & 63) == 0) slot[(threadIdx.x >> 6) * N + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
      double r = slot[k];
#pragma unroll
      for (int w = 1; w < NW; ++w) r += slot[w * N + k];
      v[k] = r;
    }
  }
};

// ---- per-system energy / gradient -----------------------------------------------------------------
// pos / grad are the system's own arrays (LDS or global), DIM doubles per atom.  Every thread walks its
// share of each term group; energy() returns the thread's partial sum, grad() accumulates with atomics.

template <int NP, int DIM, int NA>
__device__ __forceinline__ void scatter(const Dual<NP>& e, const int (&atoms)[NA], double* grad, const double scale) {
#pragma unroll
  for (int m = 0; m < NA; ++m) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double g = scale * e.d[3 * m + c];
      if (g != 0.0) atomicAdd(&grad[atoms[m] * DIM + c], g);
    }
  }
}

// Accumulator of ff_grad.h's term gradients: atomic adds into the caller's gradient array (the wave's LDS slab inside
// the fused BFGS kernel, global memory in the stand-alone gradient kernel).
template <int DIM> struct AtomicAcc {
  double* grad;
  __device__ __forceinline__ void operator()(const int atom, const ffg::V3 f) const {
    if (f.x != 0.0) atomicAdd(&grad[atom * DIM], f.x);
    if (f.y != 0.0) atomicAdd(&grad[atom * DIM + 1], f.y);
    if (f.z != 0.0) atomicAdd(&grad[atom * DIM + 2], f.z);
  }
};

template <int DIM> __device__ __forceinline__ double pair_dist2(const double* pos, const int i, const int j, const int ndim, double (&d)[4]) {
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    d[c] = (c < ndim) ? pos[i * DIM + c] - pos[j * DIM + c] : 0.0;
    s += d[c] * d[c];
  }
  return s;
}

template <int DIM> __device__ __forceinline__ void pair_push(double* grad, const int i, const int j, const int ndim, const double (&d)[4], const double f) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c < ndim) {
      const double g = f * d[c];
      atomicAdd(&grad[i * DIM + c], g);
      atomicAdd(&grad[j * DIM + c], -g);
    }
  }
}

// Pair-term loops (the O(N^2) majority: DG distances, ETK restraints, bonds, van der Waals, electrostatics) are
// latency bound when written one term at a time: index and parameter loads of a term come from global memory and only
// then can the arithmetic start (measured: 11.6 us per MMFF energy evaluation for ~10 terms per thread).  This
// helper loads PU terms' indices and parameters up front, then runs the body on each, so PU loads are in flight.
constexpr int PU = 4;
template <int NP, typename Body>
__device__ __forceinline__ void pair_terms(const Group& g, const int ms, Body&& body) {
  const int t1 = g.starts[ms + 1];
  for (int t0 = g.starts[ms] + static_cast<int>(threadIdx.x); t0 < t1; t0 += NT * PU) {
    int2   ij[PU];
    double par[PU][NP > 0 ? NP : 1];
#pragma unroll
    for (int k = 0; k < PU; ++k) {
      const int t = t0 + k * NT;
      if (t < t1) {
        ij[k] = *reinterpret_cast<const int2*>(g.idx + 2 * t);
#pragma unroll
        for (int q = 0; q < NP; ++q) par[k][q] = g.par[NP * t + q];
      }
    }
#pragma unroll
    for (int k = 0; k < PU; ++k) {
      const int t = t0 + k * NT;
      if (t < t1) body(t, ij[k].x, ij[k].y, par[k]);
    }
  }
}

// Constraint groups of the 3-D fields (MMFF: first = 7, UFF: first = 5): distance, position, angle, torsion.
template <int DIM, bool GRAD>
__device__ __forceinline__ double constraint_terms(const Batch& b, const int first, const int ms, const double* pos, double* grad) {
  const int tid = threadIdx.x;
  double    e   = 0.0;
  auto      has = [&](const int gi) { return ((b.groupMask >> gi) & 1u) && b.g[gi].starts != nullptr; };
  if (has(first)) {
    pair_terms<3>(b.g[first], ms, [&](const int, const int i, const int j, const double* p) {
      double       d[4];
      const double dist = sqrt(pair_dist2<DIM>(pos, i, j, 3, d));
      double       et, dE;
      dist_constraint(dist, p[0], p[1], p[2], et, dE);
      if constexpr (GRAD) {
        if (dE != 0.0) pair_push<DIM>(grad, i, j, 3, d, dE / (dist > 1.0e-8 ? dist : 1.0e-8));
      } else {
        e += et;
      }
    });
  }
  if (has(first + 1)) {
    const Group& g = b.g[first + 1];
    for (int t = g.starts[ms] + tid; t < g.starts[ms + 1]; t += NT) {
      const int     i = g.idx[t];
      const double* p = g.par + 5 * t;
      const double  dx = pos[i * DIM] - p[0], dy = pos[i * DIM + 1] - p[1], dz = pos[i * DIM + 2] - p[2];
      const double  dist = sqrt(dx * dx + dy * dy + dz * dz);
      double        et, dE;
      position_constraint(dist, p[3], p[4], et, dE);
      if constexpr (GRAD) {
        if (dE != 0.0) {
          const double f = dE / (dist > 1.0e-8 ? dist : 1.0e-8);
          atomicAdd(&grad[i * DIM], f * dx);
          atomicAdd(&grad[i * DIM + 1], f * dy);
          atomicAdd(&grad[i * DIM + 2], f * dz);
        }
      } else {
        e += et;
      }
    }
  }
  if (has(first + 2)) {
    const Group& g = b.g[first + 2];
    for (int t = g.starts[ms] + tid; t < g.starts[ms + 1]; t += NT) {
      const int     a[3] = {g.idx[3 * t], g.idx[3 * t + 1], g.idx[3 * t + 2]};
      const double* p    = g.par + 3 * t;
      if constexpr (GRAD) {
        using D = Dual<9>;
        scatter<9, DIM, 3>(angle_constraint_ff(Loader<D, DIM>::get(pos, a[0], 0), Loader<D, DIM>::get(pos, a[1], 1),
                                               Loader<D, DIM>::get(pos, a[2], 2), p[0], p[1], p[2]),
                           a, grad, 1.0);
      } else {
        e += angle_constraint_ff(Loader<double, DIM>::get(pos, a[0], 0), Loader<double, DIM>::get(pos, a[1], 1),
                                 Loader<double, DIM>::get(pos, a[2], 2), p[0], p[1], p[2]);
      }
    }
  }
  if (has(first + 3)) {
    const Group& g = b.g[first + 3];
    for (int t = g.starts[ms] + tid; t < g.starts[ms + 1]; t += NT) {
      const int     a[4] = {g.idx[4 * t], g.idx[4 * t + 1], g.idx[4 * t + 2], g.idx[4 * t + 3]};
      const double* p    = g.par + 3 * t;
      if constexpr (GRAD) {
        using D = Dual<12>;
        scatter<12, DIM, 4>(torsion_constraint(Loader<D, DIM>::get(pos, a[0], 0), Loader<D, DIM>::get(pos, a[1], 1),
                                               Loader<D, DIM>::get(pos, a[2], 2), Loader<D, DIM>::get(pos, a[3], 3), p[0], p[1],
                                               p[2]),
                            a, grad, 1.0);
      } else {
        e += torsion_constraint(Loader<double, DIM>::get(pos, a[0], 0), Loader<double, DIM>::get(pos, a[1], 1),
                                Loader<double, DIM>::get(pos, a[2], 2), Loader<double, DIM>::get(pos, a[3], 3), p[0], p[1], p[2]);
      }
    }
  }
  return e;
}

// ---- term batches: every group's first terms are loaded before any of them is computed --------------------------------
// One evaluation walks 3-7 term groups.  Written group after group, each group costs a full chain of dependent
// latencies (table offsets -> indices and parameters -> positions in LDS -> arithmetic) before the next one starts:
// 8.5 us per MMFF energy evaluation of a 48-atom molecule, of which < 3 us is arithmetic (profiles/r02_conformers).
// system_eval therefore first ISSUES the loads of every group's leading batch (PUK terms per thread: 4 for the O(N^2)
// pair groups, 1 for the bonded groups, which rarely have more than 256 terms) and only then computes them in issue
// order; what a group holds beyond its leading batch is walked by a batched remainder loop.
template <int NI, int NP, int PUK> struct TermBatch {
  int    a[PUK][NI];
  double p[PUK][NP > 0 ? NP : 1];
  int    first;  // this thread's first term
  int    begin;  // the group's first term of this system (for tables indexed relative to it)
  int    end;
};

template <int NI, int NP> __device__ __forceinline__ void load_term(const Group& g, const int t, int (&a)[NI], double (&p)[NP > 0 ? NP : 1]) {
  if constexpr (NI == 2) {
    const int2 v = *reinterpret_cast<const int2*>(g.idx + 2 * t);
    a[0] = v.x, a[1] = v.y;
  } else if constexpr (NI == 4) {
    const int4 v = *reinterpret_cast<const int4*>(g.idx + 4 * t);
    a[0] = v.x, a[1] = v.y, a[2] = v.z, a[3] = v.w;
  } else {
#pragma unroll
    for (int q = 0; q < NI; ++q) a[q] = g.idx[NI * t + q];
  }
#pragma unroll
  for (int q = 0; q < NP; ++q) p[q] = g.par[NP * t + q];
}

template <int NI, int NP, int PUK> __device__ __forceinline__ void load_batch(const Group& g, const int first, TermBatch<NI, NP, PUK>& tb) {
  tb.first = first;
#pragma unroll
  for (int k = 0; k < PUK; ++k) {
    const int t = first + k * NT;
    if (t < tb.end) load_term<NI, NP>(g, t, tb.a[k], tb.p[k]);
  }
}

// What an evaluation needs to know about its system besides the positions: the table row, every group's term range
// (empty when the group is masked off) and the ETK reference distances.  All of it is constant over a minimisation, so
// the fused BFGS kernel builds it ONCE and keeps it in scalar registers; evaluations then start with the term loads
// themselves instead of a chain of offset loads (two per group, 3-11 groups per evaluation).
struct TermRange {
  int begin, end;
};
struct EvalContext {
  int           ms;  // row of the term tables
  TermRange     r[12];
  const double* ref[2];
};
// Thread rotation of a group: term t of the group is taken by thread (t - begin + rot) mod NT, with rot = the number of
// terms in the groups before it.  The bonded groups have fewer terms than the workgroup has threads; unrotated, every
// one of them starts at thread 0, so wave 0 walks ALL of them one after the other (each a chain of square roots,
// divisions and an arc cosine at FP64 latency) while wave 3 has nothing but its pair terms: the evaluation takes as
// long as wave 0.  Rotated, the groups lie end to end across the waves.
__device__ __forceinline__ int group_rotation(const EvalContext& c, const int gi) {
  int rot = 0;
#pragma unroll
  for (int k = 0; k < 12; ++k)
    if (k < gi) rot += c.r[k].end - c.r[k].begin;
  return rot & (NT - 1);
}
template <int KIND> struct GroupCount {
  // MMFF: 7 term groups, 4 optional constraint groups (7..10), the optional merged non-bonded group (11)
  static constexpr int value = KIND == NVMK_FF_DG ? 3 : KIND == NVMK_FF_ETK ? 6 : KIND == NVMK_FF_MMFF ? 12 : KIND == KIND_MMFF_C ? 12
                               : KIND == NVMK_FF_UFF ? 5 : KIND == KIND_UFF_C ? 9 : 0;
};
template <int KIND> __device__ __forceinline__ EvalContext eval_context(const Batch& b, const int sys) {
  EvalContext c;
  c.ms = b.sysMol ? b.sysMol[sys] : sys;
#pragma unroll
  for (int gi = 0; gi < 12; ++gi) {
    if (gi < GroupCount<KIND>::value) {
      // branch-free: a masked-off or absent group reads the always-valid atom offsets and gets an empty range
      const bool     enabled = ((b.groupMask >> gi) & 1u) != 0u && b.g[gi].starts != nullptr;
      const int32_t* st      = enabled ? b.g[gi].starts : b.atomStarts;
      const int      lo = st[c.ms], hi = st[c.ms + 1];
      c.r[gi] = {__builtin_amdgcn_readfirstlane(lo), __builtin_amdgcn_readfirstlane(enabled ? hi : lo)};
    } else {
      c.r[gi] = {0, 0};
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) c.ref[k] = (KIND == NVMK_FF_ETK && b.ref[k]) ? b.ref[k] + b.refStarts[k][sys] : nullptr;
  return c;
}

template <int NI, int NP, int PUK>
__device__ __forceinline__ TermBatch<NI, NP, PUK> term_batch(const Group& g, const TermRange r, const int rot) {
  TermBatch<NI, NP, PUK> tb;
  tb.begin = r.begin;
  tb.end   = r.end;
  load_batch<NI, NP, PUK>(g, tb.begin + ((static_cast<int>(threadIdx.x) + NT - rot) & (NT - 1)), tb);
  return tb;
}

// body(t, a, p): term index, its atom indices, its parameters.
template <int NI, int NP, int PUK, typename Body>
__device__ __forceinline__ void run_terms(const Group& g, TermBatch<NI, NP, PUK>& tb, Body&& body) {
  while (true) {
#pragma unroll
    for (int k = 0; k < PUK; ++k) {
      const int t = tb.first + k * NT;
      if (t < tb.end) body(t, tb.a[k], tb.p[k]);
    }
    const int next = tb.first + PUK * NT;
    if (next >= tb.end) break;
    load_batch<NI, NP, PUK>(g, next, tb);
  }
}

// GRAD = false: returns the energy partial.  GRAD = true: accumulates the gradient, returns 0.
// GRAD && WITH_E (distance-geometry field only): the gradient walk also returns the energy partial — the BFGS kernel
// evaluates the first trial point of a DG line search this way, because 96 % of those trials are accepted and the
// separate gradient evaluation of the new iterate then falls away.
template <int KIND, bool GRAD, bool WITH_E = false>
__device__ __forceinline__ double system_eval(const Batch& b, const EvalContext& ctx, const int nCoords, const double* pos, double* grad,
                                               const double w0, const double w1, const int globalCoordStart) {
  constexpr int DIM = Dim<KIND>::value;
  const int     tid = threadIdx.x;
  const int     ms  = ctx.ms;
  double        e   = 0.0;
  (void)nCoords;
  (void)grad;
  (void)ms;
  (void)w0;
  (void)w1;
  (void)globalCoordStart;

  if constexpr (KIND == NVMK_FF_QUARTIC) {
    // test field of the reference's BFGS suite (tests/test_bfgs_minimizer.cu:823-860): sum (x_p - p)^4 over the
    // GLOBAL coordinate index p; w0 != 0 includes the 4th coordinate of every atom
    for (int p = tid; p < nCoords; p += NT) {
      if ((p & 3) == 3 && w0 == 0.0) continue;
      const double diff = pos[p] - static_cast<double>(globalCoordStart + p);
      if constexpr (GRAD) {
        grad[p] += 4.0 * diff * diff * diff;
      } else {
        e += diff * diff * diff * diff;
      }
    }
    return e;
  }

  [[maybe_unused]] AtomicAcc<DIM> acc{grad};
  // positions of a term's atoms for the scalar (energy) form of the angular terms
  auto at = [&](const int atom, const int slot) { return Loader<double, DIM>::get(pos, atom, slot); };
  (void)at;

  static_assert(!WITH_E || (GRAD && KIND == NVMK_FF_DG), "energy + gradient in one walk is built for the DG field");
  if constexpr (KIND == NVMK_FF_DG) {
    constexpr bool ENERGY = !GRAD || WITH_E;
    auto t0 = term_batch<2, 3, PU>(b.g[0], ctx.r[0], group_rotation(ctx, 0));
    auto t1 = term_batch<4, 2, 1>(b.g[1], ctx.r[1], group_rotation(ctx, 1));
    auto t2 = term_batch<1, 0, 1>(b.g[2], ctx.r[2], group_rotation(ctx, 2));
    // distance violations, all 4 dimensions (dist_geom_kernels_device.cuh:37-95)
    run_terms(b.g[0], t0, [&](const int, const int* a, const double* p) {
      double       d[4];
      const double d2 = pair_dist2<DIM>(pos, a[0], a[1], 4, d);
      double       et, dE;
      dist_violation(d2, p[0], p[1], p[2], et, dE);
      if constexpr (GRAD) {
        if (dE != 0.0) pair_push<DIM>(grad, a[0], a[1], 4, d, 2.0 * dE);
      }
      if constexpr (ENERGY) e += et;
    });
    // chiral volumes, weight w0 (:97-207); RDKit's gradient is half the derivative
    run_terms(b.g[1], t1, [&](const int, const int* a, const double* p) {
      if constexpr (GRAD) {
        const int aa[4] = {a[0], a[1], a[2], a[3]};
#ifdef NVMK_FF_DUAL_GRAD
        using D     = Dual<12>;
        const D vol = chiral_volume(Loader<D, DIM>::get(pos, a[0], 0), Loader<D, DIM>::get(pos, a[1], 1),
                                    Loader<D, DIM>::get(pos, a[2], 2), Loader<D, DIM>::get(pos, a[3], 3));
        scatter<12, DIM, 4>(chiral_violation(vol, p[0], p[1], w0), aa, grad, 0.5);
#else
        ffg::grad_dg_chiral<DIM>(pos, aa, p[0], p[1], w0, acc);
#endif
      }
      if constexpr (ENERGY) e += chiral_violation(chiral_volume(at(a[0], 0), at(a[1], 1), at(a[2], 2), at(a[3], 3)), p[0], p[1], w0);
    });
    // fourth dimension, weight w1 (:209-231): E = w x4^2, RDKit gradient w x4
    run_terms(b.g[2], t2, [&](const int, const int* a, const double*) {
      const double x = pos[a[0] * DIM + 3];
      if constexpr (GRAD) atomicAdd(&grad[a[0] * DIM + 3], w1 * x);
      if constexpr (ENERGY) e += w1 * x * x;
    });
    return e;
  }

  if constexpr (KIND == NVMK_FF_ETK) {
    // 1-2 / 1-3 restraints may be re-centred per system: bounds = ref +- (max - min) / 2 unless the term's 4th
    // parameter pins the table bounds (isImproperConstrained, dist_geom.h:103-110)
    const double* ref2 = ctx.ref[0];
    const double* ref3 = ctx.ref[1];
    auto t0 = term_batch<4, 0, 1>(b.g[0], ctx.r[0], group_rotation(ctx, 0));  // indices only: the 12 parameters of the (few) torsions would pin 24 registers
    auto t1 = term_batch<4, 4, 1>(b.g[1], ctx.r[1], group_rotation(ctx, 1));
    auto t2 = term_batch<2, 4, 1>(b.g[2], ctx.r[2], group_rotation(ctx, 2));
    auto t3 = term_batch<2, 4, 1>(b.g[3], ctx.r[3], group_rotation(ctx, 3));
    auto t4 = term_batch<3, 2, 1>(b.g[4], ctx.r[4], group_rotation(ctx, 4));
    auto t5 = term_batch<2, 4, PU>(b.g[5], ctx.r[5], group_rotation(ctx, 5));
    const double r2first = (ref2 && t2.first < t2.end) ? ref2[t2.first - t2.begin] : 0.0;
    const double r3first = (ref3 && t3.first < t3.end) ? ref3[t3.first - t3.begin] : 0.0;
    // flat-bottom distance restraints in 3-D: groups 2 (1-2), 3 (1-3), 5 (long range) (:368-392, :696-729)
    auto restraint = [&](const int* a, const double lo, const double hi, const double k) {
      double     d[4];
      const Root rt = root_lean(pair_dist2<DIM>(pos, a[0], a[1], 3, d));
      double     et, dE;
      dist_constraint(rt.r, lo, hi, k, et, dE);
      if constexpr (GRAD) {
        if (dE != 0.0) pair_push<DIM>(grad, a[0], a[1], 3, d, dE * fmin(rt.rinv, 1.0e8));  // dE / max(dist, 1e-8)
      } else {
        e += et;
      }
    };
    // experimental torsions: 6 force constants + 6 signs per term (:237-313, :447-575)
    run_terms(b.g[0], t0, [&](const int t, const int* a, const double*) {
      const double* fc = b.g[0].par + 12 * t;
      if constexpr (GRAD) {
        const int aa[4] = {a[0], a[1], a[2], a[3]};
#ifdef NVMK_FF_DUAL_GRAD
        using D = Dual<12>;
        bool    ok;
        const D c = cos_dihedral(Loader<D, DIM>::get(pos, a[0], 0), Loader<D, DIM>::get(pos, a[1], 1),
                                 Loader<D, DIM>::get(pos, a[2], 2), Loader<D, DIM>::get(pos, a[3], 3), ok);
        if (ok) scatter<12, DIM, 4>(torsion_m6(c, fc, fc + 6), aa, grad, 1.0);
#else
        ffg::grad_etk_torsion<DIM>(pos, aa, fc, acc);
#endif
      } else {
        bool         ok;
        const double c = cos_dihedral(at(a[0], 0), at(a[1], 1), at(a[2], 2), at(a[3], 3), ok);
        e += torsion_m6(ok ? c : 0.0, fc, fc + 6);  // degenerate: cosPhi = 0 (:286-288)
      }
    });
    // improper torsions / inversions: C0, C1, C2, k (:315-366, :577-694)
    run_terms(b.g[1], t1, [&](const int, const int* a, const double* p) {
      if constexpr (GRAD) {
        const int aa[4] = {a[0], a[1], a[2], a[3]};
#ifdef NVMK_FF_DUAL_GRAD
        using D = Dual<12>;
        scatter<12, DIM, 4>(inversion(Loader<D, DIM>::get(pos, a[0], 0), Loader<D, DIM>::get(pos, a[1], 1),
                                      Loader<D, DIM>::get(pos, a[2], 2), Loader<D, DIM>::get(pos, a[3], 3), p[0], p[1], p[2], p[3]),
                            aa, grad, 1.0);
#else
        ffg::grad_inversion<DIM>(pos, aa, p[1], p[2], p[3], false, acc);
#endif
      } else {
        e += inversion(at(a[0], 0), at(a[1], 1), at(a[2], 2), at(a[3], 3), p[0], p[1], p[2], p[3]);
      }
    });
    auto recentred = [&](const double* ref, const double refFirst, const int first, const int begin) {
      return [&, ref, refFirst, first, begin](const int t, const int* a, const double* p) {
        double lo = p[0], hi = p[1];
        if (ref && p[3] == 0.0) {
          const double centre = (t == first) ? refFirst : ref[t - begin];
          const double half   = 0.5 * (hi - lo);
          lo                  = centre - half;
          hi                  = centre + half;
        }
        restraint(a, lo, hi, p[2]);
      };
    };
    {
      const int first2 = t2.first, first3 = t3.first;  // run_terms advances .first through the remainder
      run_terms(b.g[2], t2, recentred(ref2, r2first, first2, t2.begin));
      run_terms(b.g[3], t3, recentred(ref3, r3first, first3, t3.begin));
    }
    // 1-3 angle restraints, force constant 1 (:394-445, :731-830)
    run_terms(b.g[4], t4, [&](const int, const int* a, const double* p) {
      if constexpr (GRAD) {
        const int aa[3] = {a[0], a[1], a[2]};
#ifdef NVMK_FF_DUAL_GRAD
        using D = Dual<9>;
        scatter<9, DIM, 3>(angle_constraint(Loader<D, DIM>::get(pos, a[0], 0), Loader<D, DIM>::get(pos, a[1], 1),
                                            Loader<D, DIM>::get(pos, a[2], 2), p[0], p[1], 1.0),
                           aa, grad, 1.0);
#else
        ffg::grad_angle_window<DIM>(pos, aa, p[0], p[1], 1.0, acc);
#endif
      } else {
        e += angle_constraint(at(a[0], 0), at(a[1], 1), at(a[2], 2), p[0], p[1], 1.0);
      }
    });
    // long-range restraints last (their loads are the largest)
    run_terms(b.g[5], t5, [&](const int, const int* a, const double* p) { restraint(a, p[0], p[1], p[2]); });
    return e;
  }

  if constexpr (KIND == NVMK_FF_MMFF || KIND == KIND_MMFF_C) {
    // issue order = compute order: the bonded groups first; while they compute, the larger loads of the pair groups
    // are still arriving
    auto t1 = term_batch<3, 3, 1>(b.g[1], ctx.r[1], group_rotation(ctx, 1));
    auto t2 = term_batch<3, 5, 1>(b.g[2], ctx.r[2], group_rotation(ctx, 2));
    auto t3 = term_batch<4, 1, 1>(b.g[3], ctx.r[3], group_rotation(ctx, 3));
    auto t4 = term_batch<4, 3, 1>(b.g[4], ctx.r[4], group_rotation(ctx, 4));
    auto t0 = term_batch<2, 2, 1>(b.g[0], ctx.r[0], group_rotation(ctx, 0));
    // merged non-bonded pairs (the normal case: groups 5 and 6 are then empty).  The separate tables are only walked when
    // they could not be merged or a mask selects one of them; their batches are loaded where they are used, not up here,
    // so that the common case does not hold registers for three pair batches
    auto t11 = term_batch<2, 5, PU>(b.g[11], ctx.r[11], group_rotation(ctx, 11));
    // angle bend: theta0, ka, isLinear
    run_terms(b.g[1], t1, [&](const int, const int* a, const double* p) {
      if constexpr (GRAD) {
        const int aa[3] = {a[0], a[1], a[2]};
#ifdef NVMK_FF_DUAL_GRAD
        using D = Dual<9>;
        scatter<9, DIM, 3>(mmff_angle(Loader<D, DIM>::get(pos, a[0], 0), Loader<D, DIM>::get(pos, a[1], 1),
                                      Loader<D, DIM>::get(pos, a[2], 2), p[0], p[1], p[2] != 0.0),
                           aa, grad, 1.0);
#else
        ffg::grad_mmff_angle<DIM>(pos, aa, p[0], p[1], p[2] != 0.0, acc);
#endif
      } else {
        e += mmff_angle(at(a[0], 0), at(a[1], 1), at(a[2], 2), p[0], p[1], p[2] != 0.0);
      }
    });
    // stretch-bend: theta0, r0ij, r0kj, kbaIJK, kbaKJI
    run_terms(b.g[2], t2, [&](const int, const int* a, const double* p) {
      if constexpr (GRAD) {
        const int aa[3] = {a[0], a[1], a[2]};
#ifdef NVMK_FF_DUAL_GRAD
        using D = Dual<9>;
        scatter<9, DIM, 3>(mmff_stretch_bend(Loader<D, DIM>::get(pos, a[0], 0), Loader<D, DIM>::get(pos, a[1], 1),
                                             Loader<D, DIM>::get(pos, a[2], 2), p[0], p[1], p[2], p[3], p[4]),
                           aa, grad, 1.0);
#else
        ffg::grad_mmff_stretch_bend<DIM>(pos, aa, p, acc);
#endif
      } else {
        e += mmff_stretch_bend(at(a[0], 0), at(a[1], 1), at(a[2], 2), p[0], p[1], p[2], p[3], p[4]);
      }
    });
    // out-of-plane: koop
    run_terms(b.g[3], t3, [&](const int, const int* a, const double* p) {
      if constexpr (GRAD) {
        const int aa[4] = {a[0], a[1], a[2], a[3]};
#ifdef NVMK_FF_DUAL_GRAD
        using D = Dual<12>;
        scatter<12, DIM, 4>(mmff_oop(Loader<D, DIM>::get(pos, a[0], 0), Loader<D, DIM>::get(pos, a[1], 1),
                                     Loader<D, DIM>::get(pos, a[2], 2), Loader<D, DIM>::get(pos, a[3], 3), p[0]),
                            aa, grad, 1.0);
#else
        ffg::grad_mmff_oop<DIM>(pos, aa, p[0], acc);
#endif
      } else {
        e += mmff_oop(at(a[0], 0), at(a[1], 1), at(a[2], 2), at(a[3], 3), p[0]);
      }
    });
    // torsion: V1, V2, V3
    run_terms(b.g[4], t4, [&](const int, const int* a, const double* p) {
      if constexpr (GRAD) {
        const int aa[4] = {a[0], a[1], a[2], a[3]};
#ifdef NVMK_FF_DUAL_GRAD
        using D = Dual<12>;
        bool    ok;
        const D c = cos_dihedral(Loader<D, DIM>::get(pos, a[0], 0), Loader<D, DIM>::get(pos, a[1], 1),
                                 Loader<D, DIM>::get(pos, a[2], 2), Loader<D, DIM>::get(pos, a[3], 3), ok);
        if (ok) scatter<12, DIM, 4>(mmff_torsion(c, p[0], p[1], p[2]), aa, grad, 1.0);
#else
        ffg::grad_mmff_torsion<DIM>(pos, aa, p[0], p[1], p[2], acc);
#endif
      } else {
        bool         ok;
        const double c = cos_dihedral(at(a[0], 0), at(a[1], 1), at(a[2], 2), at(a[3], 3), ok);
        e += mmff_torsion(ok ? c : 0.0, p[0], p[1], p[2]);
      }
    });
    // radial terms share their tail: distance, the term's (energy, dE/dr), force along the pair
    auto radial = [&](const int* a, auto&& term) {
      double     d[4];
      const Root rt = root_lean(pair_dist2<DIM>(pos, a[0], a[1], 3, d));
      double     et, dE;
      term(rt.r, et, dE);
      if constexpr (GRAD) {
        pair_push<DIM>(grad, a[0], a[1], 3, d, dE * rt.rinv);  // coincident atoms: d = 0, no force (as with the r > 0 test)
      } else {
        e += et;
      }
    };
    // bond stretch: r0, kb
    run_terms(b.g[0], t0, [&](const int, const int* a, const double* p) {
      radial(a, [&](const double r, double& et, double& dE) { mmff_bond(r, p[0], p[1], et, dE); });
    });
    // van der Waals: R*, eps
    auto t5 = term_batch<2, 2, PU>(b.g[5], ctx.r[5], group_rotation(ctx, 5));
    run_terms(b.g[5], t5, [&](const int, const int* a, const double* p) {
      radial(a, [&](const double r, double& et, double& dE) { mmff_vdw(r, p[0], p[1], et, dE); });
    });
    // electrostatics: chargeTerm, dielModel, is1_4
    auto t6 = term_batch<2, 3, PU>(b.g[6], ctx.r[6], group_rotation(ctx, 6));
    run_terms(b.g[6], t6, [&](const int, const int* a, const double* p) {
      radial(a, [&](const double r, double& et, double& dE) { mmff_ele(r, p[0], static_cast<int>(p[1]), p[2] != 0.0, et, dE); });
    });
    // merged non-bonded pairs: R*, eps, chargeTerm, dielModel, is1_4 — van der Waals and electrostatics of a pair share the
    // distance, its square root, the position reads and the six atomic adds (the two lists name the same pairs)
    run_terms(b.g[11], t11, [&](const int, const int* a, const double* p) {
      radial(a, [&](const double r, double& et, double& dE) {
        double ev, dv, ee, de;
        mmff_vdw(r, p[0], p[1], ev, dv);
        mmff_ele(r, p[2], static_cast<int>(p[3]), p[4] != 0.0, ee, de);
        et = ev + ee;
        dE = dv + de;
      });
    });
    if constexpr (KIND == KIND_MMFF_C) e += constraint_terms<DIM, GRAD>(b, 7, ms, pos, grad);
    return e;
  }

  if constexpr (KIND == NVMK_FF_UFF || KIND == KIND_UFF_C) {
    auto t4 = term_batch<2, 3, PU>(b.g[4], ctx.r[4], group_rotation(ctx, 4));
    auto t0 = term_batch<2, 2, 1>(b.g[0], ctx.r[0], group_rotation(ctx, 0));
    auto t1 = term_batch<3, 6, 1>(b.g[1], ctx.r[1], group_rotation(ctx, 1));
    auto t2 = term_batch<4, 3, 1>(b.g[2], ctx.r[2], group_rotation(ctx, 2));
    auto t3 = term_batch<4, 4, 1>(b.g[3], ctx.r[3], group_rotation(ctx, 3));
    const double one[4] = {1.0, 1.0, 1.0, 0.0};
    // van der Waals: x_ij, wellDepth, threshold
    run_terms(b.g[4], t4, [&](const int, const int* a, const double* p) {
      double       d[4];
      const double r = sqrt(pair_dist2<DIM>(pos, a[0], a[1], 3, d));
      double       et, dE;
      uff_vdw(r, p[0], p[1], p[2], et, dE);
      if constexpr (GRAD) {
        if (r > 0.0) {
          if (dE != 0.0) pair_push<DIM>(grad, a[0], a[1], 3, d, dE / r);
        } else if (r <= p[2]) {  // coincident atoms inside the cutoff: +-100 per component (:552-560)
          pair_push<DIM>(grad, a[0], a[1], 3, one, 100.0);
        }
      } else {
        e += et;
      }
    });
    // bond stretch: r0, k
    run_terms(b.g[0], t0, [&](const int, const int* a, const double* p) {
      double       d[4];
      const double r = sqrt(pair_dist2<DIM>(pos, a[0], a[1], 3, d));
      double       et, dE;
      uff_bond(r, p[0], p[1], et, dE);
      if constexpr (GRAD) {
        if (r > 0.0) {
          pair_push<DIM>(grad, a[0], a[1], 3, d, dE / r);
        } else {  // coincident atoms: the reference pushes them apart along (1, 1, 1) with k / 100 (:56-58)
          pair_push<DIM>(grad, a[0], a[1], 3, one, p[1] * 0.01);
        }
      } else {
        e += et;
      }
    });
    // angle bend: theta0, k, order, C0, C1, C2
    run_terms(b.g[1], t1, [&](const int, const int* a, const double* p) {
      if constexpr (GRAD) {
        const int aa[3] = {a[0], a[1], a[2]};
#ifdef NVMK_FF_DUAL_GRAD
        using D = Dual<9>;
        scatter<9, DIM, 3>(uff_angle(Loader<D, DIM>::get(pos, a[0], 0), Loader<D, DIM>::get(pos, a[1], 1),
                                     Loader<D, DIM>::get(pos, a[2], 2), p[0], p[1], static_cast<int>(p[2]), p[3], p[4], p[5]),
                           aa, grad, 1.0);
#else
        ffg::grad_uff_angle<DIM>(pos, aa, p, acc);
#endif
      } else {
        e += uff_angle(at(a[0], 0), at(a[1], 1), at(a[2], 2), p[0], p[1], static_cast<int>(p[2]), p[3], p[4], p[5]);
      }
    });
    // torsion: k, order, cosTerm
    run_terms(b.g[2], t2, [&](const int, const int* a, const double* p) {
      const int ord = static_cast<int>(p[1]);
      if constexpr (GRAD) {
        const int aa[4] = {a[0], a[1], a[2], a[3]};
#ifdef NVMK_FF_DUAL_GRAD
        using D = Dual<12>;
        bool    ok;
        const D c = cos_dihedral(Loader<D, DIM>::get(pos, a[0], 0), Loader<D, DIM>::get(pos, a[1], 1),
                                 Loader<D, DIM>::get(pos, a[2], 2), Loader<D, DIM>::get(pos, a[3], 3), ok);
        if (ok) scatter<12, DIM, 4>(uff_torsion(c, p[0], ord, p[2]), aa, grad, 1.0);
#else
        ffg::grad_uff_torsion<DIM>(pos, aa, p[0], ord, p[2], acc);
#endif
      } else {
        bool         ok;
        const double c = cos_dihedral(at(a[0], 0), at(a[1], 1), at(a[2], 2), at(a[3], 3), ok);
        e += uff_torsion(ok ? c : 0.0, p[0], ord, p[2]);  // collinear: cos(phi) := 0 (:271-273)
      }
    });
    // inversion: k, C0, C1, C2
    run_terms(b.g[3], t3, [&](const int, const int* a, const double* p) {
      if constexpr (GRAD) {
        const int aa[4] = {a[0], a[1], a[2], a[3]};
#ifdef NVMK_FF_DUAL_GRAD
        using D = Dual<12>;
        scatter<12, DIM, 4>(uff_inversion(Loader<D, DIM>::get(pos, a[0], 0), Loader<D, DIM>::get(pos, a[1], 1),
                                          Loader<D, DIM>::get(pos, a[2], 2), Loader<D, DIM>::get(pos, a[3], 3), p[0], p[1], p[2], p[3]),
                            aa, grad, 1.0);
#else
        ffg::grad_inversion<DIM>(pos, aa, p[2], p[3], p[0], true, acc);
#endif
      } else {
        e += uff_inversion(at(a[0], 0), at(a[1], 1), at(a[2], 2), at(a[3], 3), p[0], p[1], p[2], p[3]);
      }
    });
    if constexpr (KIND == KIND_UFF_C) e += constraint_terms<DIM, GRAD>(b, 5, ms, pos, grad);
    return e;
  }
  return e;
}

// ---- stand-alone energy / gradient kernels (one workgroup per system) -----------------------------
template <int KIND>
__global__ __launch_bounds__(NT) void energy_kernel(const Batch b, const double* __restrict__ pos, const double w0, const double w1,
                                                    const uint8_t* __restrict__ active, double* __restrict__ energies) {
  constexpr int DIM = Dim<KIND>::value;
  __shared__ double red[NT / 64 + 1];
  const int         sys = blockIdx.x;
  if (active && !active[sys]) return;
  const int    a0 = b.atomStarts[sys];
  const double e  = system_eval<KIND, false>(b, eval_context<KIND>(b, sys), (b.atomStarts[sys + 1] - a0) * DIM,
                                             pos + static_cast<int64_t>(a0) * DIM, nullptr, w0, w1, a0 * DIM);
  const double s  = block_reduce<Op::kSum>(e, red);
  if (threadIdx.x == 0) energies[sys] = s;
}

template <int KIND>
__global__ __launch_bounds__(NT) void grad_kernel(const Batch b, const double* __restrict__ pos, const double w0, const double w1,
                                                  const uint8_t* __restrict__ active, double* __restrict__ grad) {
  constexpr int DIM = Dim<KIND>::value;
  const int     sys = blockIdx.x;
  if (active && !active[sys]) return;
  const int a0 = b.atomStarts[sys];
  const int n  = (b.atomStarts[sys + 1] - a0) * DIM;
  double*   g  = grad + static_cast<int64_t>(a0) * DIM;
  for (int p = threadIdx.x; p < n; p += NT) g[p] = 0.0;
  __syncthreads();
  system_eval<KIND, true>(b, eval_context<KIND>(b, sys), (b.atomStarts[sys + 1] - a0) * DIM, pos + static_cast<int64_t>(a0) * DIM, g, w0, w1,
                          a0 * DIM);
}

// ---- fused BFGS -----------------------------------------------------------------------------------
constexpr double FUNCTOL       = 1.0e-4;
constexpr double MOVETOL       = 1.0e-7;
constexpr double TOLX          = 4.0 * 3.0e-8;
constexpr double EPS_HESS      = 3.0e-8;
constexpr int    MAX_LS_ITERS  = 1000;

// PROFILE (NVMK_BFGS_PROFILE=1; DG, ETK and MMFF): thread 0 accumulates wall-clock ticks (100 MHz) per phase into
// prof[sys * 8 + k]: 0 line-search energy evaluations, 1 gradient, 2 pass over H (pending update + H g), 3 update scalars + direction,
// 4 whole kernel, 5 iterations, 6 energy evaluations.
//
// Size classes (the reference switches between shared and global memory per launch, bfgs_minimize_permol_kernels.cu:796-932,
// bfgs_types.h:36-43; here every launch is split by what a system's 17 n-vectors need):
//   class A  vectors <= half the LDS of a CU: one workgroup per system, two workgroups per CU, per-system inverse Hessians;
//   class B  vectors <= the whole LDS: one workgroup per CU;
//   class C  anything larger (GVEC): the vectors live in a per-workgroup HBM / L2 work area, any size.
// Classes B and C run as persistent workgroups that take systems off a counter (largest first) and keep ONE inverse-Hessian
// slot each, so the memory a launch needs is bounded by the workgroups in flight, not by the number of large systems
// (a 1000-atom 4-D system has a 64 MB triangle).
struct BfgsArgs {
  double*                         positions;
  double                          w0, w1;
  int                             maxIters;
  double                          gradTol;
  int                             scaleGrads;
  const uint8_t*                  active;
  const int64_t*                  hessStarts;   // per-system offsets into `hessians` (slotDoubles == 0)
  const int32_t*                  order;        // the systems of this launch in hand-out order
  int                             nItems;
  int*                            counter;      // persistent launches: next item to hand out (starts at 0); else nullptr
  double*                         hessians;
  int64_t                         slotDoubles;  // > 0: workgroup k owns hessians[k * slotDoubles ...)
  double*                         vecWork;      // GVEC: workgroup k owns vecWork[k * vecStride ...)
  int64_t                         vecStride;
  double*                         energies;
  int16_t*                        statuses;
  int32_t*                        itersOut;
  int64_t*                        prof;
  int                             ldsDoubles;
  unsigned long long*             stats;
};

template <int KIND, bool GVEC, bool PROFILE>
__device__ __forceinline__ void bfgs_system(const Batch& b, const BfgsArgs& A, const int sys, char* smem) {
  double* __restrict__ const          positions  = A.positions;
  const double                        w0 = A.w0, w1 = A.w1;
  const int                           maxIters   = A.maxIters;
  const double                        gradTol    = A.gradTol;
  const int                           scaleGrads = A.scaleGrads;
  double* __restrict__ const          energies   = A.energies;
  int16_t* __restrict__ const         statuses   = A.statuses;
  int32_t* __restrict__ const         itersOut   = A.itersOut;
  int64_t* __restrict__ const         prof       = A.prof;
  unsigned long long* __restrict__ const stats   = A.stats;
  int64_t tk[7] = {0, 0, 0, 0, 0, 0, 0};
  auto    now   = [&]() -> int64_t { return PROFILE ? static_cast<int64_t>(wall_clock64()) : 0; };
  const int64_t tStart = now();
  (void)tk;
  (void)tStart;
  constexpr int DIM = Dim<KIND>::value;
  if (A.active && !A.active[sys]) return;
  const int tid = threadIdx.x;
  const int a0  = b.atomStarts[sys];
  const int n   = (b.atomStarts[sys + 1] - a0) * DIM;
  double*   gpos = positions + static_cast<int64_t>(a0) * DIM;
  double*   H    = A.hessians + (A.slotDoubles > 0 ? static_cast<int64_t>(blockIdx.x) * A.slotDoubles : A.hessStarts[sys]);

  double* pos;
  if constexpr (GVEC) {
    pos = A.vecWork + static_cast<int64_t>(blockIdx.x) * A.vecStride;
  } else {
    pos = reinterpret_cast<double*>(smem);
  }
  double* grad  = pos + n;
  double* dir   = grad + n;   // search direction, then the step actually taken (xi)
  double* trial = dir + n;    // line-search positions
  double* dGrad = trial + n;
  double* oldp  = dGrad + n;
  double* hg    = oldp + n;   // H g of the current iterate (= -direction before any rescaling)
  double* tvec  = hg + n;     // H g_new from the pass
  double* pxi   = tvec + n;   // pending rank-2 update: xi, H dGrad, u
  double* phdg  = pxi + n;
  double* pu    = phdg + n;
  double* hdiag = pu + n;     // diagonal of the inverse Hessian (the strict lower triangle is in Hl / H)
  double* part  = hdiag + n;  // (1 + NW) n partial sums of the pass; its first NW slabs double as the per-wave gradients
  // kRedDoubles of reduction scratch: always LDS
  double* redScratch;
  if constexpr (GVEC) {
    __shared__ double redStatic[kRedDoubles];
    redScratch = redStatic;
  } else {
    redScratch = part + (1 + NW) * n;
  }
  BlockReducer br{redScratch, 0};
  // Inverse Hessian: the first Rl rows of the packed triangle live in LDS behind the vectors (as many as the launch's LDS
  // budget holds: all of them for small systems), rows Rl.. stream from HBM as before.
  double*   Hl = GVEC ? nullptr : br.red + kRedDoubles;
  const int Rl = GVEC ? 0 : resident_rows(n, lds_hessian_doubles(A.ldsDoubles, n));

  if (n == 0) {
    if (tid == 0) {
      energies[sys] = 0.0;
      if (statuses) statuses[sys] = 0;
      if (itersOut) itersOut[sys] = 0;
    }
    return;
  }

  for (int i = tid; i < n; i += NT) pos[i] = gpos[i];
  {
    const int64_t nl = hess_row_offset(Rl), total = hess_row_offset(n);
    double2*      L2 = reinterpret_cast<double2*>(Hl);
    for (int64_t i = tid; i < nl / 2; i += NT) L2[i] = make_double2(0.0, 0.0);
    double2* H2 = reinterpret_cast<double2*>(H);
    for (int64_t i = tid; i < (total - nl) / 2; i += NT) H2[i] = make_double2(0.0, 0.0);
    for (int r = tid; r < n; r += NT) hdiag[r] = 1.0;  // H = identity
  }

  // table row, term ranges and reference-distance pointers of this system: loaded once, kept in LDS
  __shared__ EvalContext ctx;
  if (tid == 0) ctx = eval_context<KIND>(b, sys);
  __syncthreads();
  auto energy_at = [&](const double* p) -> double {
    return br.run<Op::kSum>(system_eval<KIND, false>(b, ctx, n, p, nullptr, w0, w1, a0 * DIM));
  };
  double gradScale = 1.0;
  // Gradient contributions are accumulated per WAVE (LDS atomics into the wave's own slab: within a wave the order of
  // the additions is the program's, so it does not depend on how the waves happen to be scheduled) and the slabs are
  // summed in a fixed order: a minimisation — and with it a seeded ETKDG run — is reproducible bit for bit.
  // `alsoMax` rides along in the gradient's own max-reduction (one barrier for both).  No barrier at the end: what follows
  // touches grad / dGrad at the thread's own indices only, up to the next reduction.
  // DG only: energy AND per-wave gradient slabs of a trial point in one walk over the terms
  auto energy_and_slabs_at = [&](const double* p) -> double {
    for (int i = tid; i < NW * n; i += NT) part[i] = 0.0;
    __syncthreads();
    double e = 0.0;
    if constexpr (KIND == NVMK_FF_DG) e = system_eval<KIND, true, true>(b, ctx, n, p, part + (tid >> 6) * n, w0, w1, a0 * DIM);
    return br.run<Op::kSum>(e);  // its barrier also completes the slabs
  };
  auto   grad_at   = [&](const double* p, double& alsoMax, const bool slabsValid) {
    if (!slabsValid) {
      for (int i = tid; i < NW * n; i += NT) part[i] = 0.0;
      __syncthreads();
      system_eval<KIND, true>(b, ctx, n, p, part + (tid >> 6) * n, w0, w1, a0 * DIM);
      __syncthreads();
    }
    // gradient scaling (bfgs_minimize_permol_kernels.cu:239-275; |g| rule of RDKit >= 2025.09)
    gradScale = scaleGrads ? 0.1 : 1.0;
    double mx = 0.0;
    for (int i = tid; i < n; i += NT) {
      double gi = part[i];
#pragma unroll
      for (int w = 1; w < NW; ++w) gi += part[w * n + i];
      if (scaleGrads) gi *= gradScale;
      grad[i] = gi;
      mx      = fmax(mx, fabs(gi));
    }
    double two[2] = {mx, alsoMax};
    br.sums_and_maxima<0, 2>(two);
    mx      = two[0];
    alsoMax = two[1];
    if (scaleGrads && mx > 10.0) {
      while (mx * gradScale > 10.0) gradScale *= 0.5;
      for (int i = tid; i < n; i += NT) grad[i] *= gradScale;
    }
  };

  // The first energy / gradient evaluation runs through the loop body as a step of length zero (`init`), so that the
  // kernel holds ONE copy of the evaluation code: inlined at two call sites each, the MMFF kernel was 108 KB of
  // instructions against a 64 KB instruction cache shared by two CUs.
  for (int i = tid; i < n; i += NT) {
    dir[i]  = 0.0;
    grad[i] = 0.0;
  }
  __syncthreads();
  double prevE   = 0.0;
  bool   pending = false;
  double pRfac = 0.0, pFad = 0.0, pFae = 0.0;
  double maxStep2 = 0.0;

  bool init      = true;
  bool converged = false;
  int  iter      = 0;
  int  nEvals    = 0;
  while (init || (!converged && iter < maxIters)) {
    // ---- line search set-up (:54-136): |dir|^2, the slope and the step test in ONE reduction; the step bound almost
    // never bites, and when it does the two quantities that depend on the rescaled direction are formed again
    double slope, test;
    {
      double three[3] = {0.0, 0.0, 0.0};
      for (int i = tid; i < n; i += NT) {
        oldp[i] = pos[i];
        three[0] += dir[i] * dir[i];
        three[1] += dir[i] * grad[i];
        three[2] = fmax(three[2], fabs(dir[i]) / fmax(fabs(pos[i]), 1.0));
      }
      br.sums_and_maxima<2, 1>(three);
      slope = three[1];
      test  = three[2];
      if (three[0] > maxStep2) {
        const double sc = sqrt(maxStep2 / three[0]);
        double       two[2] = {0.0, 0.0};
        for (int i = tid; i < n; i += NT) {
          dir[i] *= sc;
          two[0] += dir[i] * grad[i];
          two[1] = fmax(two[1], fabs(dir[i]) / fmax(fabs(pos[i]), 1.0));
        }
        br.sums_and_maxima<1, 1>(two);
        slope = two[0];
        test  = two[1];
      }
    }
    const double lambdaMin = MOVETOL / (test > 0.0 ? test : 1.0e-20);
    // ---- backtracking line search (:147-196)
    double lambda = 1.0, lambda2 = 0.0, e2 = 0.0, newE = prevE;
    bool   slabsValid = false;  // the gradient slabs in `part` belong to the trial point that gets accepted
    for (int ls = 0; ls < MAX_LS_ITERS; ++ls) {
      for (int i = tid; i < n; i += NT) trial[i] = oldp[i] + lambda * dir[i];
      __syncthreads();
      const int64_t tE = now();
      if (KIND == NVMK_FF_DG && ls == 0) {
        newE       = energy_and_slabs_at(trial);
        slabsValid = true;
      } else {
        newE       = energy_at(trial);
        slabsValid = false;
      }
      tk[0] += now() - tE;
      tk[6] += 1;
      ++nEvals;
      const double eDiff = newE - prevE;
      if (lambda < lambdaMin || eDiff <= FUNCTOL * lambda * slope) break;
      double tmp;
      if (ls == 0) {
        tmp = -slope / (2.0 * (eDiff - slope));
      } else {
        const double rhs1 = eDiff - lambda * slope;
        const double rhs2 = e2 - prevE - lambda2 * slope;
        const double a    = (rhs1 / (lambda * lambda) - rhs2 / (lambda2 * lambda2)) / (lambda - lambda2);
        const double bq   = (-lambda2 * rhs1 / (lambda * lambda) + lambda * rhs2 / (lambda2 * lambda2)) / (lambda - lambda2);
        if (a == 0.0) {
          tmp = -slope / (2.0 * bq);
        } else {
          const double disc = bq * bq - 3.0 * a * slope;
          if (disc < 0.0) {
            tmp = 0.5 * lambda;
          } else if (bq <= 0.0) {
            tmp = (-bq + sqrt(disc)) / (3.0 * a);
          } else {
            tmp = -slope / (bq + sqrt(disc));
          }
        }
        tmp = fmin(tmp, 0.5 * lambda);
      }
      lambda2 = lambda;
      e2      = newE;
      lambda  = fmax(tmp, 0.1 * lambda);
    }
    // ---- accept the step (:198-229); its TOLX test is reduced together with the gradient's maximum below (one barrier
    // less per iteration; the gradient of an iterate that turns out to be converged is computed and dropped)
    double stepTest = 0.0;
    for (int i = tid; i < n; i += NT) {
      pos[i]   = trial[i];
      dir[i]   = trial[i] - oldp[i];
      dGrad[i] = grad[i];
      stepTest = fmax(stepTest, fabs(dir[i]) / fmax(fabs(trial[i]), 1.0));
    }
    prevE = newE;  // energy of the coordinates that are returned (the reference reports the pre-step energy when
                   // TOLX fires, :680-687; the step is below 1.2e-7 relative there)
    // ---- new gradient, gradient test (:277-303)
    const int64_t tG = now();
    grad_at(pos, stepTest, slabsValid);
    tk[1] += now() - tG;
    if (!init && stepTest < TOLX) {
      converged = true;
      break;
    }
    double gTest = 0.0;
    for (int i = tid; i < n; i += NT) {
      dGrad[i] = grad[i] - dGrad[i];
      gTest    = fmax(gTest, fabs(grad[i]) * fmax(fabs(pos[i]), 1.0));
    }
    gTest = br.run<Op::kMax>(gTest) / fmax(prevE * gradScale, 1.0);
    if (init) {  // H = I: the first direction is steepest descent; the step bound of the line searches (:54-60)
      double sumsq = 0.0;
      for (int i = tid; i < n; i += NT) {
        hg[i]  = grad[i];
        dir[i] = -grad[i];
        sumsq += pos[i] * pos[i];
      }
      sumsq    = br.run<Op::kSum>(sumsq);
      maxStep2 = 1.0e4 * fmax(sumsq, static_cast<double>(n) * static_cast<double>(n));
      init     = false;
      continue;
    }
    if (gTest < gradTol) {
      converged = true;
      break;
    }
    // ---- BFGS update of the inverse Hessian, new direction (:304-407) — one pass over H, see hess_pass
    const int64_t tH = now();  // (the reduction above was a barrier: every gradient entry is visible, nobody reads `part` any more)
    hess_pass<true>(hdiag, Hl, H, Rl, n, pending, pRfac, pFad, pFae, pxi, phdg, pu, grad, part);
    hess_finish(n, part, tvec);  // H is now H_k; tvec = H_k g_new (entry i is used by thread i only: no barrier)
    tk[2] += now() - tH;
    const int64_t tU = now();
    double fac = 0.0, fae = 0.0, sumDG = 0.0, sumXi = 0.0;
    for (int i = tid; i < n; i += NT) {
      const double hd = tvec[i] - hg[i];  // H dGrad
      phdg[i]         = hd;
      pxi[i]          = dir[i];
      fac += dGrad[i] * dir[i];
      fae += dGrad[i] * hd;
      sumDG += dGrad[i] * dGrad[i];
      sumXi += dir[i] * dir[i];
    }
    {
      double four[4] = {fac, fae, sumDG, sumXi};
      br.sum_n<4>(four);
      fac   = four[0];
      fae   = four[1];
      sumDG = four[2];
      sumXi = four[3];
    }
    pending = fac > 0.0 && fac * fac > EPS_HESS * sumDG * sumXi;
    if (pending) {
      pRfac = 1.0 / fac;
      pFad  = 1.0 / fae;
      pFae  = fae;
      // u = rfac xi - fad hdg, and the three dot products with the new gradient for H_new g_new
      double dx = 0.0, dh = 0.0, du = 0.0;
      for (int i = tid; i < n; i += NT) {
        const double ui = pRfac * pxi[i] - pFad * phdg[i];
        pu[i]           = ui;
        dx += pxi[i] * grad[i];
        dh += phdg[i] * grad[i];
        du += ui * grad[i];
      }
      {
        double three[3] = {dx, dh, du};
        br.sum_n<3>(three);
        dx = three[0];
        dh = three[1];
        du = three[2];
      }
      for (int i = tid; i < n; i += NT) {
        hg[i] = tvec[i] + pRfac * dx * pxi[i] - pFad * dh * phdg[i] + pFae * du * pu[i];
      }
    } else {
      for (int i = tid; i < n; i += NT) hg[i] = tvec[i];
    }
    for (int i = tid; i < n; i += NT) dir[i] = -hg[i];  // read back by this thread only until the next reduction's barrier
    tk[3] += now() - tU;
    ++iter;
  }
  for (int i = tid; i < n; i += NT) gpos[i] = pos[i];
  if (tid == 0) {
    energies[sys] = prevE;
    if (statuses) statuses[sys] = converged ? 0 : 1;
    if (itersOut) itersOut[sys] = iter;
    if (stats) {  // nvmk_bfgs_set_stats: systems, BFGS iterations, inverse-Hessian bytes the iterations stand for, energy evaluations
      unsigned long long* st = stats + 8 * (KIND < 8 ? KIND : 7);
      atomicAdd(st + 0, 1ull);
      atomicAdd(st + 1, static_cast<unsigned long long>(iter));
      atomicAdd(st + 2, static_cast<unsigned long long>(iter) * 8ull * static_cast<unsigned long long>(hess_row_offset(n)) * 2ull);
      atomicAdd(st + 3, static_cast<unsigned long long>(nEvals));
      atomicAdd(st + 4, static_cast<unsigned long long>(iter) * 16ull * static_cast<unsigned long long>(hess_row_offset(n) - hess_row_offset(Rl)));
    }
    if constexpr (PROFILE) {
      tk[4] = now() - tStart;
      tk[5] = iter;
      for (int k = 0; k < 7; ++k) prof[static_cast<int64_t>(sys) * 8 + k] = tk[k];
    }
  }
}

// Two workgroups per CU (up to 256 VGPRs each) share the LDS.
template <int KIND, bool GVEC = false, bool PROFILE = false>
__global__ __launch_bounds__(NT, 2 * NT / 256) void bfgs_kernel(const Batch b, const BfgsArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int nextItem;
  // workgroups are handed out in launch order: the largest systems go first so that the launch does not end on a long job
  int item = blockIdx.x;
  while (item < A.nItems) {
    bfgs_system<KIND, GVEC, PROFILE>(b, A, A.order[item], smem);
    if (A.counter == nullptr) break;  // one system per workgroup
    __syncthreads();                  // the system's last reads of its vectors precede the next one's first writes
    if (threadIdx.x == 0) nextItem = static_cast<int>(gridDim.x) + atomicAdd(A.counter, 1);
    __syncthreads();
    item = nextItem;
  }
}

// ---- host side ------------------------------------------------------------------------------------

int to_batch(const nvmk_ff_batch* in, Batch& out) {
  NVMK_REQUIRE(in != nullptr, "ff: NULL batch");
  NVMK_REQUIRE(in->kind >= NVMK_FF_DG && in->kind <= NVMK_FF_UFF, "ff: unknown force-field kind %d", in->kind);
  NVMK_REQUIRE(in->n_systems >= 0, "ff: negative system count");
  NVMK_REQUIRE(in->n_systems == 0 || in->atom_starts != nullptr, "ff: NULL atom_starts");
  static const int nGroups[5] = {3, 6, 7, 0, 5};
  out.kind         = in->kind;
  out.nSystems     = in->n_systems;
  out.atomStarts   = in->atom_starts;
  out.sysMol       = in->system_mol;
  out.groupMask    = in->group_mask ? in->group_mask : 0xfffu;
  out.refStarts[0] = in->etk_ref12_starts;
  out.refStarts[1] = in->etk_ref13_starts;
  out.ref[0]       = in->etk_ref12;
  out.ref[1]       = in->etk_ref13;
  if (in->kind == NVMK_FF_MMFF || in->kind == NVMK_FF_UFF) {
    const int first = (in->kind == NVMK_FF_MMFF) ? 7 : 5;
    for (int g = first; g < first + 4; ++g) {
      if (in->groups[g].starts != nullptr) out.kind = (in->kind == NVMK_FF_MMFF) ? KIND_MMFF_C : KIND_UFF_C;
    }
  }
  if (in->kind == NVMK_FF_MMFF) {
    // groups[11] = the merged non-bonded table (optional): it replaces groups 5 and 6 when BOTH are enabled; a mask that
    // selects only one of them keeps the separate tables
    const bool merged = in->groups[11].starts != nullptr && (out.groupMask & 0x60u) == 0x60u;
    out.groupMask     = merged ? ((out.groupMask & ~0x60u) | 0x800u) : (out.groupMask & ~0x800u);
  }
  for (int g = 0; g < 12; ++g) {
    out.g[g] = {in->groups[g].starts, in->groups[g].idx, in->groups[g].par};
    if (g < nGroups[in->kind] && in->n_systems > 0) {
      NVMK_REQUIRE(in->groups[g].starts != nullptr, "ff: term group %d of kind %d has NULL starts", g, in->kind);
    }
  }
  return NVMK_OK;
}

#define NVMK_FF_DISPATCH(kind, CALL)               \
  switch (kind) {                                  \
    case NVMK_FF_DG: { constexpr int K = NVMK_FF_DG; CALL; break; }           \
    case NVMK_FF_ETK: { constexpr int K = NVMK_FF_ETK; CALL; break; }         \
    case NVMK_FF_MMFF: { constexpr int K = NVMK_FF_MMFF; CALL; break; }       \
    case NVMK_FF_UFF: { constexpr int K = NVMK_FF_UFF; CALL; break; }         \
    case KIND_MMFF_C: { constexpr int K = KIND_MMFF_C; CALL; break; }         \
    case KIND_UFF_C: { constexpr int K = KIND_UFF_C; CALL; break; }           \
    default: { constexpr int K = NVMK_FF_QUARTIC; CALL; break; }              \
  }

}  // namespace minim
}  // namespace nvmk

using namespace nvmk;
using namespace nvmk::minim;

namespace {
// device counters the BFGS kernels add to when set (nvmk_bfgs_set_stats); process-wide, off by default
std::atomic<unsigned long long*> g_stats{nullptr};

// Two highest-priority streams + their fork / join events per (host thread, device), created on first use and kept: the
// large size classes of a minimisation run on them next to class A on the caller's stream.
struct SideStreams {
  hipStream_t s[2]    = {nullptr, nullptr};
  hipEvent_t  fork    = nullptr;
  hipEvent_t  join[2] = {nullptr, nullptr};
  bool        ok      = false;
};
SideStreams* side_streams(const int dev) {
  thread_local SideStreams table[64];
  if (dev < 0 || dev >= 64) return nullptr;
  SideStreams& t = table[dev];
  if (!t.ok) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return nullptr;
    for (int k = 0; k < 2; ++k) {
      if (hipStreamCreateWithPriority(&t.s[k], hipStreamNonBlocking, greatest) != hipSuccess) return nullptr;
      if (hipEventCreateWithFlags(&t.join[k], hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    if (hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    t.ok = true;
  }
  return &t;
}
}  // namespace

extern "C" {

int nvmk_bfgs_set_stats(uint64_t* d_counters) {
  g_stats.store(reinterpret_cast<unsigned long long*>(d_counters));
  return NVMK_OK;
}

int nvmk_ff_energy(const nvmk_ff_batch* batch, double w0, double w1, const double* d_pos, const uint8_t* d_active,
                   double* d_energies, void* stream) {
  Batch b;
  int   rc = to_batch(batch, b);
  if (rc != NVMK_OK) return rc;
  if (b.nSystems == 0) return NVMK_OK;
  NVMK_REQUIRE(d_pos && d_energies, "ff energy: NULL buffer");
  NVMK_FF_DISPATCH(b.kind, hipLaunchKernelGGL(energy_kernel<K>, dim3(b.nSystems), dim3(NT), 0, as_stream(stream), b, d_pos, w0, w1,
                                              d_active, d_energies));
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

int nvmk_ff_gradient(const nvmk_ff_batch* batch, double w0, double w1, const double* d_pos, const uint8_t* d_active,
                     double* d_grad, void* stream) {
  Batch b;
  int   rc = to_batch(batch, b);
  if (rc != NVMK_OK) return rc;
  if (b.nSystems == 0) return NVMK_OK;
  NVMK_REQUIRE(d_pos && d_grad, "ff gradient: NULL buffer");
  NVMK_FF_DISPATCH(b.kind, hipLaunchKernelGGL(grad_kernel<K>, dim3(b.nSystems), dim3(NT), 0, as_stream(stream), b, d_pos, w0, w1,
                                              d_active, d_grad));
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

int nvmk_bfgs_minimize(const nvmk_ff_batch* batch, const int32_t* h_atom_starts, double w0, double w1, int max_iters,
                       double grad_tol, int scale_grads, double* d_pos, const uint8_t* d_active, double* d_energies,
                       int16_t* d_statuses, int32_t* d_iters, void* stream_) {
  Batch b;
  int   rc = to_batch(batch, b);
  if (rc != NVMK_OK) return rc;
  if (b.nSystems == 0) return NVMK_OK;
  NVMK_REQUIRE(h_atom_starts && d_pos && d_energies, "bfgs: NULL buffer");
  NVMK_REQUIRE(max_iters >= 0, "bfgs: negative iteration count");
  hipStream_t stream = as_stream(stream_);
  const int   dim    = (b.kind == NVMK_FF_DG || b.kind == NVMK_FF_QUARTIC) ? 4 : 3;
  constexpr size_t kLdsPerCu = 160 * 1024, kLdsReserve = 1024;  // static LDS of the kernel + allocation granularity
  constexpr size_t kHalf = kLdsPerCu / 2 - kLdsReserve, kFull = kLdsPerCu - kLdsReserve;
  // LDS budget of a class-A workgroup.  NVMK_BFGS_LDS: "auto" (default) = what lets two workgroups share a CU; "full" = the
  // whole 160 KiB (one workgroup per CU); "0" = vectors only (inverse Hessians entirely in HBM, the round-1 layout); a
  // number = KiB per workgroup.  (A variant compiled for three workgroups per CU — 168 VGPRs — was worth +3 % with the
  // round-1 evaluation code and -20 % with the current one, which keeps every group's leading terms in registers: removed.)
  size_t budgetA = kHalf;
  {
    const opt::Text e = opt::get(opt::kBfgsLds);
    if (e.is("full")) {
      budgetA = kFull;
    } else if (e.set() && !e.is("auto")) {
      budgetA = std::min(static_cast<size_t>(std::max(0L, e.num(0))) * 1024, kFull);
    }
  }
  const bool allGlobal = opt::get(opt::kBfgsVectors).is("global");
  const bool overlap   = !opt::get(opt::kBfgsOverlap).is("0");

  // ---- size classes
  struct Class {
    std::vector<int32_t> order;  // systems, largest first (stable)
    int                  maxN = 0;
  };
  Class cls[3];
  for (int s = 0; s < b.nSystems; ++s) {
    const int64_t n64 = static_cast<int64_t>(h_atom_starts[s + 1] - h_atom_starts[s]) * dim;
    NVMK_REQUIRE(n64 >= 0, "bfgs: atom_starts must be non-decreasing");
    NVMK_REQUIRE(n64 <= 46000, "bfgs: a system with %lld coordinates is beyond the packed triangle's 32-bit row offsets",
                 static_cast<long long>(n64));
    const size_t vecBytes = static_cast<size_t>(lds_vector_doubles(n64)) * sizeof(double);
    const int    c        = allGlobal ? 2 : vecBytes <= std::max(kHalf, budgetA) ? 0 : vecBytes <= kFull ? 1 : 2;
    cls[c].order.push_back(s);
    cls[c].maxN = std::max(cls[c].maxN, static_cast<int>(n64));
  }
  for (Class& c : cls) {
    std::stable_sort(c.order.begin(), c.order.end(), [&](const int32_t x, const int32_t y) {
      return h_atom_starts[x + 1] - h_atom_starts[x] > h_atom_starts[y + 1] - h_atom_starts[y];
    });
  }
  // XCD-aware hand-out of class A: workgroup p runs on XCD p % 8 and every XCD has its own L2.  Conformers of one molecule
  // are neighbours in `order` (same size, stable sort) and share their term tables (system_mol), so a run of kXcdGroup
  // consecutive systems goes to ONE XCD: its L2 then holds a handful of molecules' tables instead of one per resident
  // workgroup.  Chunks of 8 * kXcdGroup systems keep the sizes balanced over the XCDs.  NVMK_BFGS_XCD_GROUP=1: plain order.
  {
    const long    g         = opt::get(opt::kBfgsXcdGroup).num(16);
    const int     kXcdGroup = g >= 1 && g <= 4096 ? static_cast<int>(g) : 16;
    auto&         order     = cls[0].order;
    const int64_t n = static_cast<int64_t>(order.size()), chunk = 8LL * kXcdGroup;
    if (kXcdGroup > 1 && b.sysMol != nullptr && n >= 2 * chunk) {
      std::vector<int32_t> grouped(order.size());
      const int64_t        full = n / chunk * chunk;  // the ragged tail keeps the plain order
      for (int64_t p = 0; p < full; ++p) {
        const int64_t c = p / chunk, q = p % chunk;
        grouped[static_cast<size_t>(p)] = order[static_cast<size_t>(c * chunk + (q % 8) * kXcdGroup + q / 8)];
      }
      for (int64_t p = full; p < n; ++p) grouped[static_cast<size_t>(p)] = order[static_cast<size_t>(p)];
      order.swap(grouped);
    }
  }

  // ---- per-class launch plans
  int dev = 0, nCu = 256;
  NVMK_HIP_CHECK(hipGetDevice(&dev));
  {
    static std::atomic<int> cuCache[64] = {};
    int&                    dummy       = nCu;
    (void)dummy;
    if (dev >= 0 && dev < 64 && cuCache[dev].load() > 0) {
      nCu = cuCache[dev].load();
    } else {
      NVMK_HIP_CHECK(hipDeviceGetAttribute(&nCu, hipDeviceAttributeMultiprocessorCount, dev));
      if (nCu <= 0) nCu = 256;
      if (dev >= 0 && dev < 64) cuCache[dev].store(nCu);
    }
  }
  struct Plan {
    bool                 used = false, gvec = false;
    size_t               shmem = 0;
    int                  ldsDoubles = 0, grid = 0;
    int64_t              slotDoubles = 0, vecStride = 0;
    std::vector<int64_t> hs;  // class A: per-system offsets (indexed by system), else empty
    StreamScratch        hessMem, startsMem, orderMem, counterMem, vecMem;
  };
  Plan plan[3];
  size_t slotBytes[3] = {0, 0, 0};
  for (int c = 0; c < 3; ++c) {
    Plan& P = plan[c];
    if (cls[c].order.empty()) continue;
    P.used = true;
    P.gvec = c == 2;
    const int    maxN     = cls[c].maxN;
    const size_t vecBytes = static_cast<size_t>(lds_vector_doubles(maxN)) * sizeof(double);
    if (P.gvec) {
      P.shmem      = 0;
      P.ldsDoubles = 0;
      P.vecStride  = (lds_vector_doubles(maxN) + 1) & ~int64_t{1};
    } else {
      const size_t budget = c == 0 ? std::min(std::max(budgetA, vecBytes), kFull) : kFull;
      P.shmem      = std::min(budget, vecBytes + static_cast<size_t>(hess_row_offset(maxN)) * sizeof(double));
      P.ldsDoubles = static_cast<int>(P.shmem / sizeof(double));
    }
    if (c == 0) {
      // offsets of the HBM part of every inverse Hessian (rows Rl.. of the packed lower triangle)
      P.hs.assign(static_cast<size_t>(b.nSystems) + 1, 0);
      int64_t at = 0;
      for (const int32_t s : cls[c].order) {
        const int n  = (h_atom_starts[s + 1] - h_atom_starts[s]) * dim;
        const int rl = resident_rows(n, lds_hessian_doubles(P.ldsDoubles, n));
        P.hs[static_cast<size_t>(s)] = at;
        at += hess_row_offset(n) - hess_row_offset(rl);
      }
      P.hs[static_cast<size_t>(b.nSystems)] = at;
      P.grid                                = static_cast<int>(cls[c].order.size());
    } else {
      int64_t slot = 0;
      for (const int32_t s : cls[c].order) {
        const int n  = (h_atom_starts[s + 1] - h_atom_starts[s]) * dim;
        const int rl = P.gvec ? 0 : resident_rows(n, lds_hessian_doubles(P.ldsDoubles, n));
        slot         = std::max<int64_t>(slot, hess_row_offset(n) - hess_row_offset(rl));
      }
      P.slotDoubles = ((slot + kHessTailPadDoubles) + 1) & ~int64_t{1};
      slotBytes[c]  = static_cast<size_t>(P.slotDoubles + P.vecStride) * sizeof(double);
      P.grid        = static_cast<int>(std::min<size_t>(cls[c].order.size(), static_cast<size_t>(nCu) * (c == 1 ? 1 : 2)));
    }
  }
  // the persistent classes together take at most half of the free memory (at least one slot each)
  if (plan[1].used || plan[2].used) {
    size_t freeB = 0, totalB = 0;
    NVMK_HIP_CHECK(hipMemGetInfo(&freeB, &totalB));
    const size_t want = slotBytes[1] * static_cast<size_t>(plan[1].grid) + slotBytes[2] * static_cast<size_t>(plan[2].grid);
    if (want > freeB / 2) {
      const double f = static_cast<double>(freeB / 2) / static_cast<double>(want);
      for (int c = 1; c < 3; ++c)
        if (plan[c].used) plan[c].grid = std::max(1, static_cast<int>(plan[c].grid * f));
    }
  }
  for (int c = 0; c < 3; ++c) {
    Plan& P = plan[c];
    if (!P.used) continue;
    const auto& order = cls[c].order;
    NVMK_HIP_CHECK(P.orderMem.alloc(order.size() * sizeof(int32_t), stream));
    NVMK_HIP_CHECK(hipMemcpyAsync(P.orderMem.ptr, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    if (c == 0) {
      NVMK_HIP_CHECK(P.hessMem.alloc(static_cast<size_t>(P.hs.back() + kHessTailPadDoubles) * sizeof(double), stream));
      NVMK_HIP_CHECK(P.startsMem.alloc(P.hs.size() * sizeof(int64_t), stream));
      NVMK_HIP_CHECK(hipMemcpyAsync(P.startsMem.ptr, P.hs.data(), P.hs.size() * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    } else {
      NVMK_HIP_CHECK(P.hessMem.alloc(static_cast<size_t>(P.slotDoubles) * static_cast<size_t>(P.grid) * sizeof(double), stream));
      NVMK_HIP_CHECK(P.counterMem.alloc(sizeof(int), stream));
      NVMK_HIP_CHECK(hipMemsetAsync(P.counterMem.ptr, 0, sizeof(int), stream));
      if (P.gvec) NVMK_HIP_CHECK(P.vecMem.alloc(static_cast<size_t>(P.vecStride) * static_cast<size_t>(P.grid) * sizeof(double), stream));
    }
  }

  const bool profile = opt::get(opt::kBfgsProfile).is("1") && (b.kind == NVMK_FF_DG || b.kind == NVMK_FF_MMFF || b.kind == NVMK_FF_ETK);
  StreamScratch profMem;
  const size_t  profWords = static_cast<size_t>(b.nSystems) * 8;
  if (profile) {
    NVMK_HIP_CHECK(profMem.alloc(profWords * sizeof(int64_t), stream));
    NVMK_HIP_CHECK(hipMemsetAsync(profMem.ptr, 0, profWords * sizeof(int64_t), stream));
  }

  auto launch = [&](const int c, hipStream_t on) -> int {
    Plan&    P = plan[c];
    BfgsArgs A;
    A.positions   = d_pos;
    A.w0          = w0;
    A.w1          = w1;
    A.maxIters    = max_iters;
    A.gradTol     = grad_tol;
    A.scaleGrads  = scale_grads;
    A.active      = d_active;
    A.hessStarts  = P.startsMem.as<int64_t>();
    A.order       = P.orderMem.as<int32_t>();
    A.nItems      = static_cast<int>(cls[c].order.size());
    A.counter     = P.counterMem.as<int>();
    A.hessians    = P.hessMem.as<double>();
    A.slotDoubles = P.slotDoubles;
    A.vecWork     = P.vecMem.as<double>();
    A.vecStride   = P.vecStride;
    A.energies    = d_energies;
    A.statuses    = d_statuses;
    A.itersOut    = d_iters;
    A.prof        = profMem.as<int64_t>();
    A.ldsDoubles  = P.ldsDoubles;
    A.stats       = g_stats.load();
    auto go = [&](auto kern) -> int {
      if (P.shmem > 64 * 1024) {
        NVMK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(P.shmem)));
      }
      hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(P.grid)), dim3(NT), P.shmem, on, b, A);
      NVMK_LAUNCH_CHECK();
      return NVMK_OK;
    };
    if (profile && !P.gvec) {
      if (b.kind == NVMK_FF_DG) return go(bfgs_kernel<NVMK_FF_DG, false, true>);
      if (b.kind == NVMK_FF_ETK) return go(bfgs_kernel<NVMK_FF_ETK, false, true>);
      return go(bfgs_kernel<NVMK_FF_MMFF, false, true>);
    }
    int r = NVMK_OK;
    if (P.gvec) {
      NVMK_FF_DISPATCH(b.kind, r = go(bfgs_kernel<K, true>));
    } else {
      NVMK_FF_DISPATCH(b.kind, r = go(bfgs_kernel<K, false>));
    }
    return r;
  };

  // The large classes go first and, when class A has work too, on side streams of the highest priority: their few long
  // workgroups start at once and the many small systems fill the rest of the chip around them, instead of one class
  // waiting for the other (a 400-atom distance-geometry minimisation alone takes longer than 4000 drug-sized ones).
  const int nUsed = (plan[0].used ? 1 : 0) + (plan[1].used ? 1 : 0) + (plan[2].used ? 1 : 0);
  if (nUsed > 1 && overlap) {
    SideStreams* side = side_streams(dev);
    NVMK_REQUIRE(side != nullptr, "bfgs: could not create the side streams of device %d", dev);
    NVMK_HIP_CHECK(hipEventRecord(side->fork, stream));
    int k = 0;
    for (int c = 2; c >= 0; --c) {
      if (!plan[c].used) continue;
      if (c == 0 || k >= 2) {
        rc = launch(c, stream);
        if (rc != NVMK_OK) return rc;
        continue;
      }
      NVMK_HIP_CHECK(hipStreamWaitEvent(side->s[k], side->fork, 0));
      rc = launch(c, side->s[k]);
      if (rc != NVMK_OK) return rc;
      NVMK_HIP_CHECK(hipEventRecord(side->join[k], side->s[k]));
      ++k;
    }
    for (int j = 0; j < k; ++j) NVMK_HIP_CHECK(hipStreamWaitEvent(stream, side->join[j], 0));
  } else {
    for (int c = 2; c >= 0; --c) {
      if (!plan[c].used) continue;
      rc = launch(c, stream);
      if (rc != NVMK_OK) return rc;
    }
  }

  if (profile) {
    std::vector<int64_t> h(profWords);
    NVMK_HIP_CHECK(hipMemcpyAsync(h.data(), profMem.ptr, profWords * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    NVMK_HIP_CHECK(hipStreamSynchronize(stream));
    double  sum[7] = {0, 0, 0, 0, 0, 0, 0};
    int64_t ran = 0, longest = 0;
    for (int sI = 0; sI < b.nSystems; ++sI) {
      if (h[static_cast<size_t>(sI) * 8 + 4] == 0) continue;
      ++ran;
      longest = std::max(longest, h[static_cast<size_t>(sI) * 8 + 4]);
      for (int k = 0; k < 7; ++k) sum[k] += static_cast<double>(h[static_cast<size_t>(sI) * 8 + k]);
    }
    if (ran > 0) {
      const double us = 0.01;  // 100 MHz ticks
      std::fprintf(stderr,
                   "[nvmk bfgs profile] kind %d systems %lld: per system mean %.1f us (max %.1f us), iterations %.1f, energy evals "
                   "%.1f | per iteration: line-search energy %.1f us, gradient %.1f us, H*dGrad %.1f us, update+direction %.1f us\n",
                   b.kind, (long long)ran, sum[4] / ran * us, longest * us, sum[5] / ran, sum[6] / ran,
                   sum[0] / std::max(sum[5], 1.0) * us, sum[1] / std::max(sum[5], 1.0) * us, sum[2] / std::max(sum[5], 1.0) * us,
                   sum[3] / std::max(sum[5], 1.0) * us);
    }
    return NVMK_OK;
  }
  NVMK_HIP_CHECK(hipStreamSynchronize(stream));  // host staging (pageable) must outlive its async copies; scratch is freed in stream order
  return NVMK_OK;
}

}  // extern "C"
