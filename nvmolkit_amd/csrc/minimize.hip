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
// MI355X design: one workgroup per system — ONE wave for small systems, two or four for larger ones (the device code,
// bfgs_device.inc, is compiled once per workgroup size); positions, gradient, direction, trial positions and gradient
// difference live in LDS for the whole minimisation; all arithmetic is fp64; term tables are read straight from HBM/L2
// (they are shared by the conformers of a molecule); the inverse Hessian is the only per-system state in HBM.  Term loops
// are thread-strided over a generic table layout (see nvmolkit_amd.h): every term group is {CSR starts, interleaved local
// atom indices, interleaved double parameters}.  This file holds what is common to the workgroup sizes and the host side
// (size classes, launches).
#include <algorithm>
#include <vector>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "common.h"
#include "options.h"
#include "ff_grad.h"
#include "ff_terms.h"

#include "bfgs_common.h"

// four waves per system (every size), eight for the largest (one system per CU), two and one for the small ones (four / eight of
// them share a CU)
#define NVMK_BFGS_NS t512
#define NVMK_BFGS_THREADS 512
#include "bfgs_device.inc"
#undef NVMK_BFGS_NS
#undef NVMK_BFGS_THREADS
#define NVMK_BFGS_NS t256
#define NVMK_BFGS_THREADS 256
#include "bfgs_device.inc"
#undef NVMK_BFGS_NS
#undef NVMK_BFGS_THREADS
#define NVMK_BFGS_NS t128
#define NVMK_BFGS_THREADS 128
#include "bfgs_device.inc"
#undef NVMK_BFGS_NS
#undef NVMK_BFGS_THREADS
#define NVMK_BFGS_NS t64
#define NVMK_BFGS_THREADS 64
#include "bfgs_device.inc"
#undef NVMK_BFGS_NS
#undef NVMK_BFGS_THREADS

namespace nvmk {
namespace minim {

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

}  // namespace minim
}  // namespace nvmk

namespace nvmk {
namespace minim {
// minimize_team.hip: the kernels of the cooperative class (several workgroups per system)
int launch_team_kernel(int threads, bool profile, unsigned grid, size_t shmem, hipStream_t stream, const Batch& b, const BfgsArgs& A);
}  // namespace minim
}  // namespace nvmk

using namespace nvmk;
using namespace nvmk::minim;
using nvmk::minim::t256::hess_row_offset;
using nvmk::minim::t256::kHessTailPadDoubles;

namespace {
// device counters the BFGS kernels add to when set (nvmk_bfgs_set_stats); process-wide, off by default
std::atomic<unsigned long long*> g_stats{nullptr};

// Highest-priority streams + their fork / join events: the larger size classes of a minimisation run on them next to the
// smallest one on the caller's stream.  Sets live in a process-wide pool per device and are LEASED for the duration of one
// call (a thread_local table leaked a full set — six streams, seven events, a pinned word — with every
// short-lived host thread, and nvmk_etkdg_embed starts fresh threads per call when batches_per_gpu > 1); the pool grows to
// the number of concurrent calls per device and is kept for the life of the process.
struct SideStreams {
  static constexpr int kStreams = 6;
  hipStream_t          s[kStreams]    = {};
  hipEvent_t           fork           = nullptr;
  hipEvent_t           join[kStreams] = {};
  int*                 started        = nullptr;  // pinned host word the large classes' workgroups count themselves into
  int*                 startedDev     = nullptr;  // its device address
  int                  dev            = -1;
  void destroy() {  // a partially created set: give back what exists
    for (int k = 0; k < kStreams; ++k) {
      if (s[k]) (void)hipStreamDestroy(s[k]);
      if (join[k]) (void)hipEventDestroy(join[k]);
    }
    if (fork) (void)hipEventDestroy(fork);
    if (started) (void)hipHostFree(started);
  }
};
// One call at a time per device may have team kernels in flight: a team's workgroups wait for each other inside the launch, so two
// team launches of DIFFERENT calls (concurrent ETKDG batches, several host threads) that each got only part of their workgroups
// resident would wait for CUs the other one holds.  Inside one call the team classes follow each other safely: a later class's
// workgroups only wait for CUs, never the other way round.
std::mutex                 g_teamMutex[64];
std::mutex                 g_sideMutex;
std::vector<SideStreams*>  g_sideFree[64];
SideStreams* create_side_streams(const int dev) {
  auto* t = new SideStreams();
  t->dev  = dev;
  int  least = 0, greatest = 0;
  bool ok = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
  for (int k = 0; ok && k < SideStreams::kStreams; ++k) {
    ok = hipStreamCreateWithPriority(&t->s[k], hipStreamNonBlocking, greatest) == hipSuccess &&
         hipEventCreateWithFlags(&t->join[k], hipEventDisableTiming) == hipSuccess;
  }
  ok = ok && hipEventCreateWithFlags(&t->fork, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipHostMalloc(reinterpret_cast<void**>(&t->started), 256, hipHostMallocMapped) == hipSuccess;
  ok = ok && hipHostGetDevicePointer(reinterpret_cast<void**>(&t->startedDev), t->started, 0) == hipSuccess;
  if (!ok) {
    t->destroy();
    delete t;
    return nullptr;
  }
  return t;
}
struct SideLease {
  SideStreams* set = nullptr;
  SideStreams* take(const int dev) {
    if (set || dev < 0 || dev >= 64) return set;
    {
      const std::lock_guard<std::mutex> lock(g_sideMutex);
      if (!g_sideFree[dev].empty()) {
        set = g_sideFree[dev].back();
        g_sideFree[dev].pop_back();
      }
    }
    if (!set) set = create_side_streams(dev);
    return set;
  }
  ~SideLease() {  // every use ends with the caller's stream waiting for the side streams and the host waiting for that stream
    if (!set) return;
    const std::lock_guard<std::mutex> lock(g_sideMutex);
    g_sideFree[set->dev].push_back(set);
  }
};
}  // namespace

extern "C" {

int nvmk_bfgs_set_stats(uint64_t* d_counters) {
  g_stats.store(reinterpret_cast<unsigned long long*>(d_counters));
  return NVMK_OK;
}

int nvmk_ff_energy(const nvmk_ff_batch* batch, double w0, double w1, const double* d_pos, const uint8_t* d_active,
                   double* d_energies, void* stream) {
  NVMK_MARK_ENTRY();
  Batch b;
  int   rc = to_batch(batch, b);
  if (rc != NVMK_OK) return rc;
  if (b.nSystems == 0) return NVMK_OK;
  NVMK_REQUIRE(d_pos && d_energies, "ff energy: NULL buffer");
  NVMK_FF_DISPATCH(b.kind, hipLaunchKernelGGL(t256::energy_kernel<K>, dim3(b.nSystems), dim3(t256::NT), 0, as_stream(stream), b, d_pos, w0, w1,
                                              d_active, d_energies));
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

int nvmk_ff_gradient(const nvmk_ff_batch* batch, double w0, double w1, const double* d_pos, const uint8_t* d_active,
                     double* d_grad, void* stream) {
  NVMK_MARK_ENTRY();
  Batch b;
  int   rc = to_batch(batch, b);
  if (rc != NVMK_OK) return rc;
  if (b.nSystems == 0) return NVMK_OK;
  NVMK_REQUIRE(d_pos && d_grad, "ff gradient: NULL buffer");
  NVMK_FF_DISPATCH(b.kind, hipLaunchKernelGGL(t256::grad_kernel<K>, dim3(b.nSystems), dim3(t256::NT), 0, as_stream(stream), b, d_pos, w0, w1,
                                              d_active, d_grad));
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

int nvmk_bfgs_minimize(const nvmk_ff_batch* batch, const int32_t* h_atom_starts, double w0, double w1, int max_iters,
                       double grad_tol, int scale_grads, double* d_pos, const uint8_t* d_active, double* d_energies,
                       int16_t* d_statuses, int32_t* d_iters, void* stream_) {
  return nvmk_bfgs_minimize_repeat(batch, h_atom_starts, w0, w1, max_iters, 0, grad_tol, scale_grads, d_pos, d_active, d_energies,
                                   d_statuses, d_iters, stream_);
}

int nvmk_bfgs_minimize_repeat(const nvmk_ff_batch* batch, const int32_t* h_atom_starts, double w0, double w1, int max_iters,
                              int restarts, double grad_tol, int scale_grads, double* d_pos, const uint8_t* d_active,
                              double* d_energies, int16_t* d_statuses, int32_t* d_iters, void* stream_) {
  return nvmk_bfgs_minimize_two_stages(batch, h_atom_starts, w0, w1, max_iters, restarts, nullptr, grad_tol, scale_grads, d_pos,
                                       d_active, d_energies, d_statuses, d_iters, stream_);
}

int nvmk_bfgs_minimize_two_stages(const nvmk_ff_batch* batch, const int32_t* h_atom_starts, double w0, double w1, int max_iters,
                                  int restarts, const nvmk_bfgs_second_stage* second, double grad_tol, int scale_grads,
                                  double* d_pos, const uint8_t* d_active, double* d_energies, int16_t* d_statuses,
                                  int32_t* d_iters, void* stream_) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(restarts >= 0, "bfgs: negative restart count");
  NVMK_REQUIRE(second == nullptr || (second->max_iters >= 0 && second->restarts >= 0 && second->d_pos_between != nullptr),
               "bfgs: the second stage needs an iteration limit, a restart count and the buffer for the coordinates in between");
  Batch b;
  int   rc = to_batch(batch, b);
  if (rc != NVMK_OK) return rc;
  if (b.nSystems == 0) return NVMK_OK;
  NVMK_REQUIRE(h_atom_starts && d_pos && d_energies, "bfgs: NULL buffer");
  NVMK_REQUIRE(max_iters >= 0, "bfgs: negative iteration count");
  hipStream_t stream = as_stream(stream_);
  const int   dim    = (b.kind == NVMK_FF_DG || b.kind == NVMK_FF_QUARTIC) ? 4 : 3;
  constexpr size_t kLdsPerCu = 160 * 1024;
  // Classes = (threads per system, workgroups per CU the LDS share is sized for).  A system of up to 176 coordinates (44
  // atoms in 4-D, 58 in 3-D) is minimised by ONE WAVE: eight such systems share a CU, every phase of a minimisation is a
  // longer loop of the same wave instead of a short one followed by a barrier (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES 0.29-0.34
  // against 0.21-0.26 with four waves), and the chip's 256 MB Infinity Cache holds the inverse Hessians of everything in
  // flight.  Larger systems keep four waves: their passes over the inverse Hessian need the parallelism, and a system that
  // takes three times longer holds up its whole launch — measured on 10 000 molecules x 10 conformers (mol/s, one box per
  // line of thresholds): 112: 2337, 144: 2543, 160: 2709, 200: 2739 | 176: 2835, 232: 2657, 300: 2454, 400: 2398; four
  // waves for everything: 2310 (profiles/r03_conformers/wave_threshold.jsonl).  Four-wave systems take two workgroups per
  // CU, one, or — vectors in HBM — any size.  Within a thread count a system goes into the first bin whose LDS share
  // holds its vectors; what the share leaves is filled with rows of its inverse Hessian.
  struct BinDef {
    int threads, wgPerCu, maxN;
  };
  // (the six-per-CU bin serves NVMK_BFGS_WAVE thresholds above the default: at 192 coordinates the vector units are
  // saturated with six systems per CU, 9.0 us per iteration and CU slot against 9.4 with eight)
  // NVMK_BFGS_WAVE: 0 = four waves for every system; a number > 1 = the largest system (coordinates) one wave takes.
  constexpr int kNoLimit = 1 << 30;
  const long    waveOpt  = opt::get(opt::kBfgsWave).num(1);
  const int     kWaveMaxN = waveOpt > 1 ? static_cast<int>(std::min<long>(waveOpt, 2000)) : 176;
  // NVMK_BFGS_WAVE2: the largest system TWO waves take (0 = none), default 256 coordinates: four such systems per CU, between
  // the one-wave and the four-wave kernels in both respects — measured (10 000 molecules x 10 conformers, mol/s; one-wave
  // threshold, two-wave threshold): (176, none) 2872, (176, 208) 2935, (176, 240) 2933, (176, 256) 2937, (176, 288) 2912,
  // (176, 336) 2837, (176, 400) 2859, (144, 256) 2821, (160, 256) 2883, (192, 256) 2832, (128, 232) 2756
  // (profiles/r03_conformers/wave2_threshold.jsonl).
  const long    wave2Opt  = opt::get(opt::kBfgsWave2).num(1);
  const int     kWave2MaxN = wave2Opt > 1 ? static_cast<int>(std::min<long>(wave2Opt, 2000)) : wave2Opt == 0 ? 0 : 256;
  const BinDef  kBins[]  = {{64, 8, 176}, {64, 6, 232}, {64, 4, kNoLimit}, {64, 3, kNoLimit}, {64, 2, kNoLimit}, {64, 1, kNoLimit},
                            {128, 4, kNoLimit}, {128, 2, kNoLimit}, {128, 1, kNoLimit},
                            {256, 2, kNoLimit}, {256, 1, kNoLimit}, {512, 1, kNoLimit}};
  constexpr int nBins = 12, kFirst128 = 6, kFirst256 = 9, kFirst512 = 11;
  // NVMK_BFGS_WAVE8: the smallest system (coordinates) EIGHT waves take (0 = none), default 656 — where the vectors of a four-wave
  // system stop fitting half a CU's LDS, so that it would have a CU to itself anyway: with four waves such a system streams its
  // inverse Hessian at ~15 GB/s, the rate four waves keep in flight, and a batch's large molecules END the batch
  // (profiles/r05_conformers/bfgs_timeline_summary_chembl_up_to_256_atoms.json: the one-per-CU class spends half of its wall with
  // fewer than 256 systems in flight).  Eight waves fill the CU's second wave slot per SIMD; their 8 gradient slabs fit LDS up to
  // 1067 coordinates, beyond that the vectors live in HBM (eight waves as well).
  const long    wave8Opt  = opt::get(opt::kBfgsWave8).num(1);
  const int     kWave8MinN = wave8Opt > 1 ? static_cast<int>(std::min<long>(wave8Opt, kNoLimit)) : wave8Opt == 0 ? kNoLimit : 656;
  // static LDS of the kernel + the 512-byte allocation granularity, per workgroup
  auto bin_budget = [&](const int c) { return kLdsPerCu / static_cast<size_t>(kBins[c].wgPerCu) - (kBins[c].wgPerCu > 2 ? 512 : 1024); };
  auto vec_doubles = [](const int threads, const int64_t n) {
    return threads == 64 ? t64::lds_vector_doubles(n) : threads == 128 ? t128::lds_vector_doubles(n) : threads == 256 ? t256::lds_vector_doubles(n) : t512::lds_vector_doubles(n);
  };
  auto hess_doubles = [](const int threads, const int64_t ldsDoubles, const int64_t n) {
    return threads == 64    ? t64::lds_hessian_doubles(ldsDoubles, n)
           : threads == 128 ? t128::lds_hessian_doubles(ldsDoubles, n)
           : threads == 256 ? t256::lds_hessian_doubles(ldsDoubles, n)
                            : t512::lds_hessian_doubles(ldsDoubles, n);
  };
  // (rows resident in LDS: the one-wave kernels round a boundary below 64 rows to a multiple of 8, see hess_pass.h)
  auto resident = [](const int threads, const int n, const int64_t hld) {
    return threads == 64 ? t64::resident_rows(n, hld) : threads == 128 ? t128::resident_rows(n, hld) : threads == 256 ? t256::resident_rows(n, hld) : t512::resident_rows(n, hld);
  };
  const size_t kFull = bin_budget(nBins - 1);
  // NVMK_BFGS_LDS: "auto" (default) = the bins above; "full" = every system gets the whole 160 KiB (one workgroup per CU);
  // "0" = vectors only (inverse Hessians entirely in HBM, the round-1 layout); a number = KiB of the first bin of a thread
  // count.  None of them changes which thread count a system gets, so none of them changes a bit of the results.
  // NVMK_BFGS_WAVE=0: four waves for every system.  (A variant compiled for three 256-thread workgroups per CU — 168 VGPRs
  // — was worth +3 % with the round-1 evaluation code and -20 % with the current one: removed.)
  bool   vectorsOnly = false, fullOnly = false;
  size_t firstBudget = 0;
  {
    const opt::Text e = opt::get(opt::kBfgsLds);
    if (e.is("full")) {
      fullOnly = true;
    } else if (e.is("0")) {
      vectorsOnly = true;
    } else if (e.set() && !e.is("auto")) {
      firstBudget = std::min(static_cast<size_t>(std::max(1L, e.num(0))) * 1024, kFull);
    }
  }
  const bool waveClass = !opt::get(opt::kBfgsWave).is("0") && b.kind != NVMK_FF_QUARTIC;
  const bool allGlobal = opt::get(opt::kBfgsVectors).is("global");
  const bool overlap   = !opt::get(opt::kBfgsOverlap).is("0");
  // NVMK_BFGS_SCHED: "queue" (default) = every class runs as persistent workgroups that take systems off per-XCD queues,
  // largest first; "hw" = one workgroup per system, handed out by the hardware dispatcher (rounds 1-3).  With "hw" a freed
  // wave slot goes to whichever class has the SMALLEST workgroups — a one-wave workgroup fits anywhere, a four-wave one needs
  // half a CU at once — so the large classes starve as soon as the small ones are on the chip and a launch group ends with
  // the LONGEST systems running alone (profiles/r04_conformers/bfgs_timeline_summary_before.json: 263 four-wave systems of a
  // 16 384-attempt batch start only when the two-wave class is exhausted, at 135 of 151 ms).  Persistent workgroups keep their
  // slots until their class is drained, so the classes finish largest first and a launch ends on the short systems.
  const bool queueMode = !opt::get(opt::kBfgsSched).is("hw");
  auto       budget_of = [&](const int c) {
    return (c == 0 || c == kFirst128 || c == kFirst256 || c == kFirst512) ? std::max(firstBudget, bin_budget(c)) : bin_budget(c);
  };
  const int  kGlobal = nBins, kGlobal8 = nBins + 1;  // class indices of the HBM-vector systems: four waves, eight waves
  // Team classes (several workgroups per system, minimize_team.hip): class kTeam0 + k = teams of 2^k workgroups.  A system of
  // NVMK_BFGS_TEAM coordinates or more (default 656: the systems the eight-wave one-workgroup class used to take) joins the
  // class of the largest power-of-two width that leaves every rank NVMK_BFGS_TEAM_SHARE_KB of the packed inverse Hessian (at
  // most one XCD's CUs); NVMK_BFGS_TEAM_WIDTH sets one width for all of them (tests: any number up to
  // 256, ranks then count across the XCDs).  The width of a system's team depends on its size only, and so do its results.
  constexpr int kTeam0 = 14, kTeamClasses = 9, kNumClasses = kTeam0 + kTeamClasses;
  const long    teamOpt   = opt::get(opt::kBfgsTeam).num(-1);
  const int     kTeamMinN = teamOpt == 0 ? kNoLimit : teamOpt > 0 ? static_cast<int>(std::min<long>(teamOpt, kNoLimit)) : 656;
  const long    teamWidthOpt = opt::get(opt::kBfgsTeamWidth).num(0);
  const long    teamShareKb  = std::max<long>(1, opt::get(opt::kBfgsTeamShareKb).num(4096));
  const int     teamThreads  = opt::get(opt::kBfgsTeamThreads).num(512) == 256 ? 256 : 512;
  // a rank's LDS (trial positions + gradient slabs / the pass's staging area, bfgs_device.inc): all of a CU's for one workgroup of
  // 512 threads, half of it for each of two workgroups of 256; a system must leave room for one gradient slab behind its positions
  const int     teamLdsDoubles = teamThreads == 512 ? 19200 : 9600;
  // History form of a team system's inverse Hessian (bfgs_device.inc: history_product_held): NVMK_BFGS_HISTORY auto (default: a
  // system for which three times the call's iteration limit is at most twice its coordinates — every team system of an ETKDG or
  // MMFF run) | 0 | 1 (every team system).  A launch that uses it sets three doubles per pair a rank may own aside at the end of
  // its LDS.  Whole ChEMBL file, ten conformers: ETKDG 35.3 -> 16.3 s, MMFF 5.1 -> 4.0 s (profiles/r06_conformers/history_form.txt).
  const opt::Text historyOpt  = opt::get(opt::kBfgsHistory);
  const int       historyK    = historyOpt.is("0") ? 0 : std::max(max_iters, second ? second->max_iters : 0);
  const bool      historyAll  = historyOpt.is("1");
  const bool      historyMay  = historyK > 0 && historyK <= kHistOwnedCap * 256;
  const int     kTeamMaxN      = (teamLdsDoubles - (historyMay ? 3 * std::min(historyK, kHistOwnedCap) : 0)) / 2;
  // class kTeam0 + k: teams of 2^k workgroups.  A "team" of ONE is the same kernel without an exchange (NVMK_BFGS_TEAM_WIDTH=1);
  // the library itself never picks it: with the history form the whole ChEMBL file takes ETKDG 18.8 / MMFF 4.0 s with teams of
  // one for the systems below 2 x NVMK_BFGS_TEAM_SHARE_KB, 18.0 / 4.0 s with teams of two (profiles/r06_conformers/history_form_sweeps.txt).
  const int     minWidthLog2 = 1;
  int           teamWidth[kTeamClasses];
  for (int k = 0; k < kTeamClasses; ++k) teamWidth[k] = 1 << k;
  auto team_class_of = [&](const int64_t n64) -> int {
    if (teamWidthOpt > 0) {
      const int w = static_cast<int>(std::min<long>(teamWidthOpt, 256));
      int       k = 0;
      while (k + 1 < kTeamClasses && (1 << k) < w) ++k;
      teamWidth[k] = std::max(w, 1);
      return kTeam0 + k;
    }
    const int64_t bytes = hess_row_offset(n64) * 8;
    int           k     = 0;
    while (k + 1 < 6 && bytes / (2 << k) >= teamShareKb * 1024) ++k;  // widths 1 .. 32
    return kTeam0 + std::max(k, minWidthLog2);
  };

  // ---- size classes
  struct Class {
    std::vector<int32_t> order;  // systems, largest first (stable)
    int                  maxN = 0;
  };
  Class cls[kNumClasses];
  for (int s = 0; s < b.nSystems; ++s) {
    const int64_t n64 = static_cast<int64_t>(h_atom_starts[s + 1] - h_atom_starts[s]) * dim;
    NVMK_REQUIRE(n64 >= 0, "bfgs: atom_starts must be non-decreasing");
    NVMK_REQUIRE(n64 <= 46000, "bfgs: a system with %lld coordinates is beyond the packed triangle's 32-bit row offsets",
                 static_cast<long long>(n64));
    const bool wave8 = waveClass && n64 >= kWave8MinN;
    int        c     = wave8 ? kGlobal8 : kGlobal;
    if (!allGlobal && n64 >= kTeamMinN && n64 <= kTeamMaxN) {
      c = team_class_of(n64);
    } else if (!allGlobal) {
      // thread count by size alone, then the bin by the LDS policy
      const bool wave  = waveClass && n64 <= kWaveMaxN;
      const bool wave2 = !wave && waveClass && n64 <= kWave2MaxN;
      const int  lo = wave ? 0 : wave2 ? kFirst128 : wave8 ? kFirst512 : kFirst256, hi = wave ? kFirst128 : wave2 ? kFirst256 : wave8 ? nBins : kFirst512;
      for (int k = fullOnly ? hi - 1 : lo; k < hi; ++k)
        if (n64 <= kBins[k].maxN && static_cast<size_t>(vec_doubles(kBins[k].threads, n64)) * sizeof(double) <= budget_of(k)) {
          c = k;
          break;
        }
    }
    cls[c].order.push_back(s);
    cls[c].maxN = std::max(cls[c].maxN, static_cast<int>(n64));
  }
  for (Class& c : cls) {
    std::stable_sort(c.order.begin(), c.order.end(), [&](const int32_t x, const int32_t y) {
      return h_atom_starts[x + 1] - h_atom_starts[x] > h_atom_starts[y + 1] - h_atom_starts[y];
    });
  }
  // XCD-aware hand-out of the one-system-per-workgroup bins: workgroup p runs on XCD p % 8 and every XCD has its own L2.
  // Conformers of one molecule are neighbours in `order` (same size, stable sort) and share their term tables
  // (system_mol), so a run of kXcdGroup consecutive systems goes to ONE XCD: its L2 then holds a handful of molecules'
  // tables instead of one per resident workgroup.  Chunks of 8 * kXcdGroup systems keep the sizes balanced over the XCDs.
  // NVMK_BFGS_XCD_GROUP=1: plain order.
  auto persistent = [&](const int c) { return queueMode || c >= kGlobal || kBins[c].wgPerCu == 1; };
  for (int c = 0; c < nBins; ++c) {
    if (persistent(c)) continue;
    const long    g         = opt::get(opt::kBfgsXcdGroup).num(32);
    const int     kXcdGroup = g >= 1 && g <= 4096 ? static_cast<int>(g) : 32;
    auto&         order     = cls[c].order;
    const int64_t n = static_cast<int64_t>(order.size()), chunk = 8LL * kXcdGroup;
    if (kXcdGroup > 1 && b.sysMol != nullptr && n >= 2 * chunk) {
      std::vector<int32_t> grouped(order.size());
      const int64_t        full = n / chunk * chunk;  // the ragged tail keeps the plain order
      for (int64_t p = 0; p < full; ++p) {
        const int64_t cc = p / chunk, q = p % chunk;
        grouped[static_cast<size_t>(p)] = order[static_cast<size_t>(cc * chunk + (q % 8) * kXcdGroup + q / 8)];
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
    if (dev >= 0 && dev < 64 && cuCache[dev].load() > 0) {
      nCu = cuCache[dev].load();
    } else {
      NVMK_HIP_CHECK(hipDeviceGetAttribute(&nCu, hipDeviceAttributeMultiprocessorCount, dev));
      if (nCu <= 0) nCu = 256;
      if (dev >= 0 && dev < 64) cuCache[dev].store(nCu);
    }
  }
  struct Plan {
    bool                 used = false, gvec = false, persistent = false;
    int                  threads = 256;
    size_t               shmem = 0;
    int                  ldsDoubles = 0, grid = 0;
    int64_t              slotDoubles = 0, vecStride = 0;
    int                  queueStart[9] = {};
    bool                 oneQueue = false;
    std::vector<int64_t> hs;  // one-system-per-workgroup bins: per-system offsets (indexed by system), else empty
    StreamScratch        hessMem, startsMem, orderMem, counterMem, vecMem;
    // team classes: workgroups per team, teams in the launch, the exchange area and control words of the teams
    int                             teamSize = 0, nTeams = 0, historyPairs = 0;
    int64_t                         exchStride = 0, teamVecStride = 0;
    StreamScratch                   exchMem, ctrlMem;
    std::vector<unsigned long long> ctrlHost;
  };
  Plan   plan[kNumClasses];
  size_t slotBytes[kNumClasses] = {};
  for (int c = 0; c <= kGlobal8; ++c) {
    Plan& P = plan[c];
    if (cls[c].order.empty()) continue;
    P.used       = true;
    P.gvec       = c >= kGlobal;
    P.persistent = persistent(c);
    P.threads    = c == kGlobal8 ? 512 : P.gvec ? 256 : kBins[c].threads;
    const int    maxN     = cls[c].maxN;
    const size_t vecBytes = static_cast<size_t>(vec_doubles(P.threads, maxN)) * sizeof(double);
    if (P.gvec) {
      P.shmem      = 0;
      P.ldsDoubles = 0;
      P.vecStride  = (vec_doubles(P.threads, maxN) + 1) & ~int64_t{1};
    } else {
      const size_t budget = vectorsOnly ? vecBytes : budget_of(c);
      P.shmem      = std::min(budget, vecBytes + static_cast<size_t>(hess_row_offset(maxN)) * sizeof(double));
      P.ldsDoubles = static_cast<int>(P.shmem / sizeof(double));
    }
    if (!P.persistent) {
      // offsets of the HBM part of every inverse Hessian (rows Rl.. of the packed lower triangle)
      P.hs.assign(static_cast<size_t>(b.nSystems) + 1, 0);
      int64_t at = 0;
      for (const int32_t s : cls[c].order) {
        const int n  = (h_atom_starts[s + 1] - h_atom_starts[s]) * dim;
        const int rl = resident(P.threads, n, hess_doubles(P.threads, P.ldsDoubles, n));
        P.hs[static_cast<size_t>(s)] = at;
        at += hess_row_offset(n) - hess_row_offset(rl);
      }
      P.hs[static_cast<size_t>(b.nSystems)] = at;
      P.grid                                = static_cast<int>(cls[c].order.size());
    } else {
      int64_t slot = 0;
      for (const int32_t s : cls[c].order) {
        const int n  = (h_atom_starts[s + 1] - h_atom_starts[s]) * dim;
        const int rl = P.gvec ? 0 : resident(P.threads, n, hess_doubles(P.threads, P.ldsDoubles, n));
        slot         = std::max<int64_t>(slot, hess_row_offset(n) - hess_row_offset(rl));
      }
      P.slotDoubles = ((slot + kHessTailPadDoubles) + 1) & ~int64_t{1};
      slotBytes[c]  = static_cast<size_t>(P.slotDoubles + P.vecStride) * sizeof(double);
      P.grid        = static_cast<int>(std::min<size_t>(cls[c].order.size(), static_cast<size_t>(nCu) * (P.gvec ? (P.threads == 512 ? 1 : 2) : kBins[c].wgPerCu)));
    }
  }
  // Team classes: persistent teams of `width` workgroups, one workgroup per CU (512 threads) or two (256); the workgroups form
  // teams in the order they start (bfgs_device.inc: make_team).  Memory: one slot for the largest system's packed triangle per TEAM, the HBM vectors per workgroup, the
  // exchange area (two blocks of `width` partial vectors, the reduced vector, two rows of scalars) and eight control words per team.
  for (int c = kTeam0; c < kNumClasses; ++c) {
    Plan& P = plan[c];
    if (cls[c].order.empty()) continue;
    P.used       = true;
    P.gvec       = true;
    P.persistent = true;
    P.threads    = teamThreads;
    const int maxN     = cls[c].maxN;
    const int capacity = nCu * (teamThreads == 512 ? 1 : 2);  // workgroups that are resident together
    int       width    = std::max(1, std::min(teamWidth[c - kTeam0], capacity));
    const int nItems   = static_cast<int>(cls[c].order.size());
    P.nTeams = std::max(1, std::min(capacity / width, nItems));
    P.teamSize = width;
    P.grid     = P.nTeams * width;
    P.ldsDoubles        = teamLdsDoubles;
    P.shmem             = static_cast<size_t>(P.ldsDoubles) * sizeof(double);
    P.vecStride         = (vec_doubles(P.threads, maxN) + 1) & ~int64_t{1};
    // the team's slot: the largest packed triangle among the systems that keep one, or width x the records of a rank's share of
    // the pairs for the largest system in the history form (a rank's part of the slot: slotDoubles / width)
    const int ownedCap = historyMay ? (historyK + width - 1) / width : 0;
    P.historyPairs     = (historyMay && ownedCap <= kHistOwnedCap) ? historyK : 0;
    int64_t slot       = 0;
    for (const int32_t s : cls[c].order) {
      const int64_t n = static_cast<int64_t>(h_atom_starts[s + 1] - h_atom_starts[s]) * dim;
      if (P.historyPairs > 0 && (historyAll || 3 * static_cast<int64_t>(P.historyPairs) <= 2 * n) && n <= 16 * teamThreads) {
        // (+ 128 doubles per rank: a DMA chunk reads up to 127 doubles past a vector's end)
        slot = std::max<int64_t>(slot, static_cast<int64_t>(width) * (static_cast<int64_t>(ownedCap) * 2 * ((n + 1) & ~int64_t{1}) + 128));
      } else {
        slot = std::max<int64_t>(slot, hess_row_offset(n) + kHessTailPadDoubles);
      }
    }
    P.slotDoubles       = (slot + 2 * width - 1) / (2 * width) * (2 * width);
    P.teamVecStride     = (static_cast<int64_t>(maxN) + 2 + 1) & ~int64_t{1};
    P.exchStride        = ((2 * static_cast<int64_t>(width) + 1) * P.teamVecStride + 2 * width + 1) & ~int64_t{1};
    for (int q = 0; q <= 8; ++q) P.queueStart[q] = q == 0 ? 0 : nItems;
    P.oneQueue = true;
  }
  // A one-system-per-workgroup class holds the HBM part of EVERY system's inverse Hessian for the whole launch (16 384
  // attempts of ~150-atom molecules: 20-30 GB).  A class that wants more than a quarter of the free memory runs as a
  // persistent class instead: as many workgroups as fit on the chip, one slot each, systems taken off a counter — same
  // kernel, same results, memory bounded by the workgroups in flight.
  {
    size_t freeB = 0, totalB = 0;
    bool   asked = false;
    for (int c = 0; c < nBins; ++c) {
      Plan& P = plan[c];
      if (!P.used || P.persistent) continue;
      const size_t want = static_cast<size_t>(P.hs.back() + kHessTailPadDoubles) * sizeof(double);
      const long   capOpt = opt::get(opt::kBfgsHessCapMb).num(0);  // NVMK_BFGS_HESS_CAP_MB (tests): the limit in MiB instead of free / 4
      if (capOpt <= 0 && want < (size_t{1} << 30)) continue;       // below 1 GiB: not worth a query
      if (!asked && capOpt <= 0) {
        NVMK_HIP_CHECK(hipMemGetInfo(&freeB, &totalB));
        asked = true;
      }
      const size_t limit = capOpt > 0 ? static_cast<size_t>(capOpt) << 20 : freeB / 4;
      if (want <= limit) continue;
      int64_t slot = 0;
      for (const int32_t s : cls[c].order) {
        const int n  = (h_atom_starts[s + 1] - h_atom_starts[s]) * dim;
        const int rl = resident(P.threads, n, hess_doubles(P.threads, P.ldsDoubles, n));
        slot         = std::max<int64_t>(slot, hess_row_offset(n) - hess_row_offset(rl));
      }
      P.persistent  = true;
      P.oneQueue    = true;  // its order is already arranged for the hardware hand-out
      P.hs.clear();
      P.slotDoubles = ((slot + kHessTailPadDoubles) + 1) & ~int64_t{1};
      slotBytes[c]  = static_cast<size_t>(P.slotDoubles) * sizeof(double);
      P.grid        = static_cast<int>(std::min<size_t>(cls[c].order.size(), static_cast<size_t>(nCu) * kBins[c].wgPerCu));
    }
  }
  // the persistent classes together take at most half of the free memory (at least one slot each)
  {
    size_t want = 0;
    for (int c = 0; c <= kGlobal8; ++c)
      if (plan[c].used && plan[c].persistent) want += slotBytes[c] * static_cast<size_t>(plan[c].grid);
    if (want > 0) {
      size_t freeB = 0, totalB = 0;
      NVMK_HIP_CHECK(hipMemGetInfo(&freeB, &totalB));
      if (want > freeB / 2) {
        const double f = static_cast<double>(freeB / 2) / static_cast<double>(want);
        for (int c = 0; c <= kGlobal8; ++c)
          if (plan[c].used && plan[c].persistent) plan[c].grid = std::max(1, static_cast<int>(plan[c].grid * f));
      }
    }
  }
  // Queues of the persistent classes: runs of kXcdGroup consecutive systems (conformers of one molecule are neighbours in
  // `order` and share their term tables) are dealt round-robin to eight queues, one per XCD — a workgroup takes from the
  // queue of the XCD it runs on (its L2 then holds a handful of molecules' tables) and from the others' once that is empty.
  // Every queue keeps the largest-first order.  Few items, or no shared tables: one queue.
  for (int c = 0; c <= kGlobal8; ++c) {
    Plan& P = plan[c];
    if (!P.used || !P.persistent) continue;
    auto&         order = cls[c].order;
    const int64_t n     = static_cast<int64_t>(order.size());
    const long    g     = opt::get(opt::kBfgsXcdGroup).num(32);
    const int     kXcdGroup = g >= 1 && g <= 4096 ? static_cast<int>(g) : 32;
    for (int q = 0; q <= 8; ++q) P.queueStart[q] = q == 0 ? 0 : static_cast<int>(n);
    if (P.oneQueue || b.sysMol == nullptr || kXcdGroup <= 1 || n < 16LL * kXcdGroup) continue;  // (team classes: one queue)
    std::vector<int32_t> queued;
    queued.reserve(order.size());
    for (int q = 0; q < 8; ++q) {
      P.queueStart[q] = static_cast<int>(queued.size());
      for (int64_t run = q; run * kXcdGroup < n; run += 8)
        for (int64_t p = run * kXcdGroup; p < std::min<int64_t>(n, (run + 1) * kXcdGroup); ++p) queued.push_back(order[static_cast<size_t>(p)]);
    }
    P.queueStart[8] = static_cast<int>(queued.size());
    order.swap(queued);
  }
  for (int c = 0; c < kNumClasses; ++c) {
    Plan& P = plan[c];
    if (!P.used) continue;
    const auto& order = cls[c].order;
    NVMK_HIP_CHECK(P.orderMem.alloc(order.size() * sizeof(int32_t), stream));
    NVMK_HIP_CHECK(hipMemcpyAsync(P.orderMem.ptr, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    if (P.teamSize > 0) {
      NVMK_HIP_CHECK(P.hessMem.alloc(static_cast<size_t>(P.slotDoubles) * static_cast<size_t>(P.nTeams) * sizeof(double), stream));
      NVMK_HIP_CHECK(P.vecMem.alloc(static_cast<size_t>(P.vecStride) * static_cast<size_t>(P.grid) * sizeof(double), stream));
      NVMK_HIP_CHECK(P.exchMem.alloc(static_cast<size_t>(P.exchStride) * static_cast<size_t>(P.nTeams) * sizeof(double), stream));
      // (the teams' control words, then one word of start tickets)
      NVMK_HIP_CHECK(P.ctrlMem.alloc((static_cast<size_t>(P.nTeams) * kTeamCtrlWords + 1) * sizeof(unsigned long long), stream));
      NVMK_HIP_CHECK(hipMemsetAsync(P.ctrlMem.ptr, 0, (static_cast<size_t>(P.nTeams) * kTeamCtrlWords + 1) * sizeof(unsigned long long), stream));
      NVMK_HIP_CHECK(P.counterMem.alloc(9 * sizeof(int), stream));
      NVMK_HIP_CHECK(hipMemsetAsync(P.counterMem.ptr, 0, 9 * sizeof(int), stream));
    } else if (!P.persistent) {
      NVMK_HIP_CHECK(P.hessMem.alloc(static_cast<size_t>(P.hs.back() + kHessTailPadDoubles) * sizeof(double), stream));
      NVMK_HIP_CHECK(P.startsMem.alloc(P.hs.size() * sizeof(int64_t), stream));
      NVMK_HIP_CHECK(hipMemcpyAsync(P.startsMem.ptr, P.hs.data(), P.hs.size() * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    } else {
      NVMK_HIP_CHECK(P.hessMem.alloc(static_cast<size_t>(P.slotDoubles) * static_cast<size_t>(P.grid) * sizeof(double), stream));
      NVMK_HIP_CHECK(P.counterMem.alloc(9 * sizeof(int), stream));  // eight queue counters + the number of items taken
      NVMK_HIP_CHECK(hipMemsetAsync(P.counterMem.ptr, 0, 9 * sizeof(int), stream));
      if (P.gvec) NVMK_HIP_CHECK(P.vecMem.alloc(static_cast<size_t>(P.vecStride) * static_cast<size_t>(P.grid) * sizeof(double), stream));
    }
  }

  const bool profile = opt::get(opt::kBfgsProfile).is("1") && (b.kind == NVMK_FF_DG || b.kind == NVMK_FF_MMFF || b.kind == NVMK_FF_ETK);
  StreamScratch profMem;
  const size_t  profWords = static_cast<size_t>(b.nSystems) * kProfWords;
  if (profile) {
    NVMK_HIP_CHECK(profMem.alloc(profWords * sizeof(int64_t), stream));
    NVMK_HIP_CHECK(hipMemsetAsync(profMem.ptr, 0, profWords * sizeof(int64_t), stream));
  }

  std::unique_lock<std::mutex> teamLock;  // held from here to the end of the call when it has team classes
  for (int c = kTeam0; c < kNumClasses; ++c) {
    if (plan[c].used && !teamLock.owns_lock() && dev >= 0 && dev < 64) teamLock = std::unique_lock<std::mutex>(g_teamMutex[dev]);
  }
  int*      startedDev = nullptr;  // set when several classes run side by side (see below)
  SideLease sideLease;            // returned to the pool when this call ends (it ends with a stream synchronisation)
  auto launch = [&](const int c, hipStream_t on) -> int {
    Plan&    P = plan[c];
    BfgsArgs A;
    A.positions   = d_pos;
    A.w0          = w0;
    A.w1          = w1;
    A.maxIters    = max_iters;
    A.restarts    = restarts;
    A.w0b         = second ? second->w0 : 0.0;
    A.w1b         = second ? second->w1 : 0.0;
    A.maxItersB   = second ? second->max_iters : -1;
    A.restartsB   = second ? second->restarts : 0;
    A.posMid      = second ? second->d_pos_between : nullptr;
    A.skipAbove   = second ? second->skip_above_energy_per_atom : -1.0;
    A.gradTol     = grad_tol;
    A.scaleGrads  = scale_grads;
    A.active      = d_active;
    A.hessStarts  = P.startsMem.as<int64_t>();
    A.order       = P.orderMem.as<int32_t>();
    A.nItems      = static_cast<int>(cls[c].order.size());
    A.counter     = P.counterMem.as<int>();
    for (int q = 0; q <= 8; ++q) A.queueStart[q] = P.queueStart[q];
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
    A.started     = (on != stream) ? startedDev : nullptr;
    A.drained     = (startedDev != nullptr && P.persistent) ? startedDev + 1 + c : nullptr;
    A.teamSize       = P.teamSize;
    A.teamTickets    = reinterpret_cast<unsigned*>(P.ctrlMem.as<unsigned long long>() + static_cast<size_t>(P.nTeams) * kTeamCtrlWords);
    A.teamExchange   = P.exchMem.as<double>();
    A.teamExchStride = P.exchStride;
    A.teamVecStride  = P.teamVecStride;
    A.teamCtrl       = P.ctrlMem.as<unsigned long long>();
    A.teamTimeout    = std::max<long>(1, opt::get(opt::kBfgsTeamTimeoutMs).num(60000)) * 100000LL;  // 100 MHz ticks
    A.historyPairs   = P.historyPairs;
    A.historyForce   = historyAll ? 1 : 0;
    A.historyOwned   = P.historyPairs > 0 ? (P.historyPairs + std::max(P.teamSize, 1) - 1) / std::max(P.teamSize, 1) : 0;
    char label[96];
    std::snprintf(label, sizeof(label), "BFGS %s: %d systems x %d threads%s", b.kind == NVMK_FF_DG ? "DG" : b.kind == NVMK_FF_ETK ? "ETK"
                  : (b.kind == NVMK_FF_MMFF || b.kind == KIND_MMFF_C) ? "MMFF" : (b.kind == NVMK_FF_UFF || b.kind == KIND_UFF_C) ? "UFF" : "quartic",
                  A.nItems, P.threads, P.gvec ? " (vectors in HBM)" : "");
    if (P.teamSize > 0) {
      std::snprintf(label, sizeof(label), "BFGS team class: %d systems, %d teams x %d workgroups x %d threads", A.nItems, P.nTeams, P.teamSize, P.threads);
    }
    NVMK_MARK(label);  // the launch of this size class
    if (P.teamSize > 0) return launch_team_kernel(P.threads, profile, static_cast<unsigned>(P.grid), P.shmem, on, b, A);
    auto go = [&](auto kern) -> int {
      if (P.shmem > 64 * 1024) {
        NVMK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(P.shmem)));
      }
      hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(P.grid)), dim3(static_cast<unsigned>(P.threads)), P.shmem, on, b, A);
      NVMK_LAUNCH_CHECK();
      return NVMK_OK;
    };
    if (profile && !P.gvec) {
      if (P.threads == 64) {
        if (b.kind == NVMK_FF_DG) return go(t64::bfgs_kernel<NVMK_FF_DG, false, true>);
        if (b.kind == NVMK_FF_ETK) return go(t64::bfgs_kernel<NVMK_FF_ETK, false, true>);
        return go(t64::bfgs_kernel<NVMK_FF_MMFF, false, true>);
      }
      if (P.threads == 128) {
        if (b.kind == NVMK_FF_DG) return go(t128::bfgs_kernel<NVMK_FF_DG, false, true>);
        if (b.kind == NVMK_FF_ETK) return go(t128::bfgs_kernel<NVMK_FF_ETK, false, true>);
        return go(t128::bfgs_kernel<NVMK_FF_MMFF, false, true>);
      }
      if (P.threads == 512) {
        if (b.kind == NVMK_FF_DG) return go(t512::bfgs_kernel<NVMK_FF_DG, false, true>);
        if (b.kind == NVMK_FF_ETK) return go(t512::bfgs_kernel<NVMK_FF_ETK, false, true>);
        return go(t512::bfgs_kernel<NVMK_FF_MMFF, false, true>);
      }
      if (b.kind == NVMK_FF_DG) return go(t256::bfgs_kernel<NVMK_FF_DG, false, true>);
      if (b.kind == NVMK_FF_ETK) return go(t256::bfgs_kernel<NVMK_FF_ETK, false, true>);
      return go(t256::bfgs_kernel<NVMK_FF_MMFF, false, true>);
    }
    int r = NVMK_OK;
    if (P.gvec && P.threads == 512) {
      NVMK_FF_DISPATCH(b.kind, r = go(t512::bfgs_kernel<K, true>));
    } else if (P.gvec) {
      NVMK_FF_DISPATCH(b.kind, r = go(t256::bfgs_kernel<K, true>));
    } else if (P.threads == 512) {
      NVMK_FF_DISPATCH(b.kind, r = go(t512::bfgs_kernel<K, false>));
    } else if (P.threads == 64) {
      NVMK_FF_DISPATCH(b.kind, r = go(t64::bfgs_kernel<K, false>));
    } else if (P.threads == 128) {
      NVMK_FF_DISPATCH(b.kind, r = go(t128::bfgs_kernel<K, false>));
    } else {
      NVMK_FF_DISPATCH(b.kind, r = go(t256::bfgs_kernel<K, false>));
    }
    return r;
  };

  // The classes of the large systems go first and every class but the last runs on a side stream of the highest priority:
  // the few long workgroups of the large classes start at once and the many small systems fill the rest of the chip around
  // them, instead of one class waiting for the other (a 400-atom distance-geometry minimisation alone takes longer than
  // 4000 drug-sized ones).
  int nUsed = 0, lastUsed = -1;
  for (int c = 0; c < kNumClasses; ++c)
    if (plan[c].used) {
      ++nUsed;
      if (lastUsed < 0) lastUsed = c;  // the bin of the smallest systems in use stays on the caller's stream
    }
  if (nUsed > 1 && overlap) {
    SideStreams* side = sideLease.take(dev);
    NVMK_REQUIRE(side != nullptr, "bfgs: could not create the side streams of device %d", dev);
    startedDev                                      = side->startedDev;
    for (int w = 0; w < 64; ++w) static_cast<volatile int*>(side->started)[w] = 0;  // [0] started, [1 + c] class c drained
    NVMK_HIP_CHECK(hipEventRecord(side->fork, stream));
    int k = 0, bigWorkgroups = 0, prevClass = -1;
    for (int c = kNumClasses - 1; c >= 0; --c) {
      if (!plan[c].used) continue;
      if (queueMode && prevClass >= 0 && plan[prevClass].persistent) {
        // Classes one after the other, WITHOUT waiting for a class to finish: the next (smaller) class is launched when the
        // last system of the previous one has been taken off its queues, so its workgroups fill the slots the previous class
        // frees while its last systems run.  Launched together, the classes interleave on a CU and the smallest one's LDS
        // blocks fragment the allocation: once the one-wave class is through, 3.3 instead of 4 two-wave workgroups fit a CU
        // for the rest of the launch (profiles/r04_conformers/bfgs_timeline_summary_queue.json: 1698 of 2048 waves).
        const auto t0 = std::chrono::steady_clock::now();
        while (static_cast<volatile int*>(side->started)[1 + prevClass] == 0 &&
               std::chrono::steady_clock::now() - t0 < std::chrono::seconds(20)) {
          std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        if (static_cast<volatile int*>(side->started)[1 + prevClass] == 0) {  // (never seen; the launches below are correct either way)
          std::fprintf(stderr, "[nvmk bfgs] size class %d had not handed out its last system after 20 s: launching the next class beside it\n", prevClass);
        }
      }
      prevClass = c;
      if (c == lastUsed || k >= SideStreams::kStreams) {
        // The larger classes must be ON the chip before the small one is launched: a workgroup of theirs needs more LDS
        // (and, with four waves, a slot on every SIMD of a CU) than a finishing small one sets free, so once the small class
        // has filled the CUs the large ones starve until its grid is exhausted — measured: the 1 % of a batch that needs
        // four waves took 127 ms next to a 76 ms launch of the rest.  Their workgroups count themselves into a pinned word;
        // the host waits (at most half a millisecond) until as many have started as can be resident.
        const int  target = queueMode ? 0 : std::min(bigWorkgroups, nCu);
        const auto t0     = std::chrono::steady_clock::now();
        while (*static_cast<volatile int*>(side->started) < target &&
               std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(500)) {
        }
        rc = launch(c, stream);
        if (rc != NVMK_OK) return rc;
        continue;
      }
      NVMK_HIP_CHECK(hipStreamWaitEvent(side->s[k], side->fork, 0));
      rc = launch(c, side->s[k]);
      if (rc != NVMK_OK) return rc;
      bigWorkgroups += plan[c].grid;
      NVMK_HIP_CHECK(hipEventRecord(side->join[k], side->s[k]));
      ++k;
    }
    for (int j = 0; j < k; ++j) NVMK_HIP_CHECK(hipStreamWaitEvent(stream, side->join[j], 0));
  } else {
    for (int c = kNumClasses - 1; c >= 0; --c) {
      if (!plan[c].used) continue;
      rc = launch(c, stream);
      if (rc != NVMK_OK) return rc;
    }
  }
  // the teams' failure flags (a barrier that gave up) come back with the results
  for (int c = kTeam0; c < kNumClasses; ++c) {
    Plan& P = plan[c];
    if (!P.used) continue;
    P.ctrlHost.assign(static_cast<size_t>(P.nTeams) * kTeamCtrlWords, 0ull);
    NVMK_HIP_CHECK(hipMemcpyAsync(P.ctrlHost.data(), P.ctrlMem.ptr, P.ctrlHost.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
  }
  auto teams_ok = [&]() -> int {
    for (int c = kTeam0; c < kNumClasses; ++c)
      for (int t = 0; plan[c].used && t < plan[c].nTeams; ++t)
        NVMK_REQUIRE(plan[c].ctrlHost[static_cast<size_t>(t) * kTeamCtrlWords + 1] == 0ull,
                     "bfgs: a team of %d workgroups gave up waiting for its members (NVMK_BFGS_TEAM_TIMEOUT_MS); are %d workgroups of %d threads resident together on this device?",
                     plan[c].teamSize, plan[c].grid, plan[c].threads);
    return NVMK_OK;
  };

  if (profile) {
    std::vector<int64_t> h(profWords);
    NVMK_HIP_CHECK(hipMemcpyAsync(h.data(), profMem.ptr, profWords * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    NVMK_HIP_CHECK(hipStreamSynchronize(stream));
    double  sum[7] = {0, 0, 0, 0, 0, 0, 0}, hist[4] = {0, 0, 0, 0};
    int64_t ran = 0, longest = 0;
    for (int sI = 0; sI < b.nSystems; ++sI) {
      if (h[static_cast<size_t>(sI) * kProfWords + 4] == 0) continue;
      ++ran;
      longest = std::max(longest, h[static_cast<size_t>(sI) * kProfWords + 4]);
      for (int k = 0; k < 7; ++k) sum[k] += static_cast<double>(h[static_cast<size_t>(sI) * kProfWords + k]);
      for (int k = 0; k < 4; ++k) hist[k] += static_cast<double>(h[static_cast<size_t>(sI) * kProfWords + 12 + k]);
    }
    // NVMK_BFGS_TIMELINE=path: one line per system that ran — kind, coordinates, first / last clock of its workgroup (100 MHz,
    // chip-wide), XCC and CU — appended; tools/bfgs_timeline.py turns the file into occupancy over time and the launch tails
    {
      const opt::Text path = opt::get(opt::kBfgsTimeline);
      if (path.set()) {
        if (std::FILE* f = std::fopen(path.s, "a")) {
          for (int sI = 0; sI < b.nSystems; ++sI) {
            const int64_t* r = h.data() + static_cast<size_t>(sI) * kProfWords;
            if (r[8] == 0) continue;
            const unsigned hw = static_cast<unsigned>(r[9] & 0xffffffff), xcc = static_cast<unsigned>(r[9] >> 32) & 0xf;
            std::fprintf(f, "%d %d %lld %lld %u %u %u %lld %lld %lld %lld %lld %lld %lld %lld\n", b.kind, (h_atom_starts[sI + 1] - h_atom_starts[sI]) * dim,
                         (long long)r[7], (long long)r[8], xcc, (hw >> 8) & 0xf, (hw >> 4) & 0x3, (long long)r[5], (long long)r[4],
                         (long long)r[10],  // (workgroups of the system's team, 0 = one workgroup)
                         (long long)r[0], (long long)r[1], (long long)r[2], (long long)r[3], (long long)r[6]);  // phase ticks: line-search energies, gradients, H g, update + direction; energy evaluations
          }
          std::fclose(f);
        }
      }
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
    if (hist[3] > 0) {
      std::fprintf(stderr, "[nvmk bfgs profile] kind %d history product: %.0f batches, per batch: loads + dot products %.2f us, reductions + coefficients %.2f us, terms %.2f us\n",
                   b.kind, hist[3], hist[0] / hist[3] * 0.01, hist[1] / hist[3] * 0.01, hist[2] / hist[3] * 0.01);
    }
    return teams_ok();
  }
  NVMK_HIP_CHECK(hipStreamSynchronize(stream));  // host staging (pageable) must outlive its async copies; scratch is freed in stream order
  return teams_ok();
}

}  // extern "C"
