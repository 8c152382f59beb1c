// Tuning / test switches of the library (the NVMK_* names of DESIGN.md).  Each is read from the environment ONCE per
// process, the first time any switch is looked up; afterwards a value changes only through nvmk_set_option (C ABI), under a
// lock.  Entry points take a consistent snapshot per call, so concurrent callers never race with getenv / setenv.
#pragma once

#include <cstdlib>
#include <cstring>

namespace nvmk {
namespace opt {

enum Id {
  kSimPath,         // NVMK_SIM_PATH         auto | mfma | valu
  kCountThreshold,  // NVMK_COUNT_THRESHOLD  (unset) | table
  kCountSuper,      // NVMK_COUNT_SUPER      supertile edge of the count kernel (experiments)
  kCountKernel,     // NVMK_COUNT_KERNEL     auto | tile | panel (symmetric all-pairs pass: 128 x 128 tiles, or row panels with the A operand in registers)
  kButinaRounds,    // NVMK_BUTINA_ROUNDS    (unset) | dense | serial
  kButinaSort,      // NVMK_BUTINA_SORT      (unset) | 0
  kBfgsLds,         // NVMK_BFGS_LDS         auto | 0 | full | KiB
  kBfgsXcdGroup,    // NVMK_BFGS_XCD_GROUP   32 | n
  kBfgsProfile,     // NVMK_BFGS_PROFILE     1
  kBfgsVectors,     // NVMK_BFGS_VECTORS     auto | global (tests: every system through the HBM-vector kernels)
  kBfgsOverlap,     // NVMK_BFGS_OVERLAP     1 | 0 (0: size classes run one after the other on the caller's stream)
  kBfgsWave,        // NVMK_BFGS_WAVE        1 | 0 | n (0: four waves for every system; n: largest system one wave takes)
  kBfgsWave2,       // NVMK_BFGS_WAVE2       n (largest system, in coordinates, that two waves take; 0: none)
  kBfgsWave8,       // NVMK_BFGS_WAVE8       n (smallest system, in coordinates, that eight waves take; 0: none; default 656)
  kBfgsHessCapMb,   // NVMK_BFGS_HESS_CAP_MB n (tests: inverse-Hessian memory of a one-system-per-workgroup class before it runs persistent; default free / 4)
  kBfgsTimeline,    // NVMK_BFGS_TIMELINE    path (with NVMK_BFGS_PROFILE=1: per-system start / end clocks appended to this file)
  kBfgsSched,       // NVMK_BFGS_SCHED       queue | hw (hw: one workgroup per system, hardware hand-out — rounds 1-3)
  kMarkers,         // NVMK_MARKERS          1 | 0 (0: no roctx ranges)
  kEtkdgTiming,     // NVMK_ETKDG_TIMING     1 (per-stage wall clock of nvmk_etkdg_embed: a stream synchronisation after every stage)
  kEtkdgPrune,      // NVMK_ETKDG_PRUNE      1 | 0 (0: surplus attempts of a molecule also run the second half of the pipeline)
  kBuildSlotKb,     // NVMK_BUILD_SLOT_KB    n (tests: size of a pinned staging slot of the table builder, default 32768)
  kBfgsTeam,        // NVMK_BFGS_TEAM        n (smallest system, in coordinates, minimised by a TEAM of workgroups; 0: none; default 656)
  kBfgsTeamWidth,   // NVMK_BFGS_TEAM_WIDTH  w (workgroups per team for every team system; default: by size, see NVMK_BFGS_TEAM_SHARE_KB)
  kBfgsTeamShareKb, // NVMK_BFGS_TEAM_SHARE_KB  k (a team has the largest power-of-two width that leaves every rank at least k KiB of the packed inverse Hessian)
  kBfgsTeamThreads, // NVMK_BFGS_TEAM_THREADS 512 | 256 (threads of a team's workgroups: one or two workgroups per CU)
  kBfgsTeamTimeoutMs, // NVMK_BFGS_TEAM_TIMEOUT_MS  t (a team barrier gives up after t ms and the call reports an error; default 60000)
  kBfgsHistory,     // NVMK_BFGS_HISTORY     auto | 0 | 1 (a team system's inverse Hessian as the pairs of its updates: where three times the iteration limit is at most twice its coordinates / never / wherever the pairs' scalars fit)
  kNumOptions
};

struct Text {
  char s[128];
  bool set() const { return s[0] != '\0'; }
  bool is(const char* v) const { return std::strcmp(s, v) == 0; }
  long num(const long dflt) const { return set() ? std::atol(s) : dflt; }
};

const char* name(Id id);
Text        get(Id id);
int         set(const char* name, const char* value);  // 0 = ok, -1 = unknown name; value NULL / "" = unset
int         find(const char* name);                     // Id or -1

}  // namespace opt
}  // namespace nvmk
