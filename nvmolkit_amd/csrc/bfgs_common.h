// Declarations shared by the translation units of the fused BFGS kernels (minimize.hip: one workgroup per system;
// minimize_team.hip: several workgroups per system): the term-table view, the kinds, RDKit's BFGS constants and the kernel
// arguments.  Included before bfgs_device.inc, which is compiled once per workgroup size.
#pragma once

#include <cstdint>

#include "common.h"
#include "ff_grad.h"
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
// bfgs_types.h:36-43; here every launch is split by size, nvmk_bfgs_minimize_two_stages below):
//   one wave per system up to 176 coordinates (eight per CU), two waves up to 256 (four per CU), four waves beyond —
//   vectors in LDS while they fit half (two workgroups per CU) or all of it (one), else (GVEC) in a per-workgroup HBM / L2
//   work area, any size.
// The one-workgroup-per-CU and the GVEC classes run as persistent workgroups that take systems off a counter (largest
// first) and keep ONE inverse-Hessian slot each, so the memory a launch needs is bounded by the workgroups in flight, not by
// the number of large systems (a 1000-atom 4-D system has a 64 MB triangle).
constexpr int kProfWords = 16;  // per system: 7 phase sums, then the item's first / last clock, its hardware id and team width (timeline); 12..15: history product — ticks in the batches' loads + dot products, reductions + coefficients, terms; batches
struct BfgsArgs {
  double*                         positions;
  double                          w0, w1;
  int                             maxIters;
  int                             restarts;     // further minimisations of a system that stops at maxIters (each from H = I)
  // optional second minimisation of every system in the same launch (maxItersB < 0: none)
  double                          w0b, w1b;
  int                             maxItersB, restartsB;
  double*                         posMid;       // coordinates after the first minimisation (same layout as positions)
  double                          skipAbove;    // >= 0: no second minimisation when the first one's energy per atom exceeds it
  double                          gradTol;
  int                             scaleGrads;
  const uint8_t*                  active;
  const int64_t*                  hessStarts;   // per-system offsets into `hessians` (slotDoubles == 0)
  const int32_t*                  order;        // the systems of this launch in hand-out order
  int                             nItems;
  int*                            counter;      // persistent launches: eight counters, one per queue (all start at 0); else nullptr
  int                             queueStart[9]; // persistent launches: queue q holds order[queueStart[q] .. queueStart[q + 1]) — one queue per XCD
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
  int*                            started;      // host-visible counter of workgroups that have begun (NULL: not wanted)
  int*                            drained;      // host-visible flag, set when the LAST item of this launch's queues has been taken (NULL: not wanted)
  // Cooperative class (bfgs_team_kernel): teamSize workgroups minimise ONE system together — the rows of its inverse Hessian and
  // its force-field terms are dealt over them, the O(n) vector work is replicated, sums cross the team through teamExchange.
  int                             teamSize;       // workgroups per system (0: not a team launch)
  unsigned*                       teamTickets;    // one word, zeroed before the launch: workgroups draw their (team, rank) in the order they start
  double*                         teamExchange;   // per team: teamExchStride doubles — teamSize partial vectors, the reduced vector, two rows of scalars
  int64_t                         teamExchStride;
  int64_t                         teamVecStride;  // doubles per vector of the exchange area (>= the largest system's coordinates + 1, even)
  unsigned long long*             teamCtrl;       // per team: kTeamCtrlWords words, zeroed before the launch (arrivals, failure flag, two item slots)
  long long                       teamTimeout;    // wall-clock ticks (100 MHz) a team barrier may wait before the launch gives up
  // The inverse Hessian of a team's system kept as its HISTORY (bfgs_device.inc: history_product): the (xi, H dGrad) pairs of the
  // rank-2 updates instead of the packed triangle they add up to.  historyPairs = pairs a minimisation may store (the launch's
  // largest iteration limit), 0 = every system keeps the triangle; a system takes the history form when 3 historyPairs <= 2 x its
  // coordinates (the pairs of a minimisation that runs to the limit then average 3/8 of the triangle's read + write per
  // iteration, and hold 4/3 of it at the very end; measured on the reference's benchmark file: the form wins from 656
  // coordinates on at 400 iterations, not from 400) or historyForce is set — and at most 16 x the workgroup's threads.
  int                             historyPairs;
  int                             historyForce;
  int                             historyOwned;   // pairs a rank may own: ceil(historyPairs / teamSize) — 3 doubles each at the end of the dynamic LDS
};
constexpr int kTeamCtrlWords = 8;
// History form: the three scalars of every pair a rank owns live at the end of the launch's dynamic LDS (3 x historyOwned doubles):
// a rank owns every teamSize-th pair, and no launch asks for more than kHistOwnedCap of them per rank.
constexpr int kHistOwnedCap = 512;

int to_batch(const nvmk_ff_batch* in, Batch& out);  // minimize.hip

}  // namespace minim
}  // namespace nvmk


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

