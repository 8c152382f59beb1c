// Batched ETKDG conformer embedding: scheduler, stage pipeline and stereochemistry checks — gfx950.
//
// Replaces (reference paths):
//   src/etkdg_impl.{h,cpp}:223-326      Scheduler (dispatch / record)
//   src/etkdg_impl.cpp:111-159          ETKDGDriver::iterate (stage loop, failure collection)
//   src/etkdg_kernels.cu:20-70          setRunFilter / collectAndFilterFailures / getFinished kernels
//   src/etkdg_stage_coordgen.cu:83-127  random 4-D start coordinates
//   src/etkdg_stage_distgeom_minimize.cu:177-249, src/etkdg_stage_etk_minimization.cu:204-266  minimisation stages
//   src/etkdg_stage_stereochem_checks.cu:25-442  tetrahedral / chiral / double-bond checks
//   src/etkdg.cpp:331-419               stage order of embedMolecules
//   src/conformer/etkdg_device_collect.cu  4-D -> 3-D packing of accepted conformers
//
// Inputs are FLATTENED per-molecule term tables: the bounds matrix, chiral sets and experimental
// torsions are RDKit's and arrive as DG / ETK term groups plus a list of stereo checks.  All conformers of a
// molecule share one copy of its tables on the device (nvmk_ff_batch.system_mol); a batch is described only by
// its atom offsets and molecule ids.  Start coordinates come from a counter-based generator on the device
// (the reference draws them on the host from RDKit's global RNG and copies them over: parity with RDKit is
// statistical on this path).
#include <algorithm>
#include <cstring>
#include <mutex>
#include <thread>
#include <string>
#include <atomic>
#include <chrono>
#include <stdexcept>
#include <vector>

#include "common.h"
#include "options.h"

namespace nvmk {
namespace etkdg {

// ---- scheduler (src/etkdg_impl.cpp:272-326) -------------------------------------------------------------
class Scheduler {
 public:
  Scheduler(int nMols, int confsPerMol, int maxIterations)
      : confs_(confsPerMol), maxTries_(maxIterations * confsPerMol), completed_(static_cast<size_t>(nMols), 0),
        attempts_(static_cast<size_t>(nMols), 0) {}

  // `attemptBase` (optional) receives the number of attempts handed out before this call: a unique, scheduler-ordered
  // id range for the random-coordinate generator
  std::vector<int> dispatch(int batchSize, uint64_t* attemptBase = nullptr) {
    std::vector<int>            ids;
    const std::lock_guard<std::mutex> lock(mutex_);
    if (attemptBase) *attemptBase = dispatched_;
    size_t                      prev = 1;
    while (static_cast<int>(ids.size()) < batchSize && prev != ids.size()) {
      prev            = ids.size();
      const int limit = std::min(maxTries_, confs_ * round_);
      for (size_t m = 0; m < completed_.size(); ++m) {
        while (completed_[m] < confs_ && attempts_[m] < limit) {
          if (static_cast<int>(ids.size()) >= batchSize) break;
          ids.push_back(static_cast<int>(m));
          ++attempts_[m];
        }
      }
      if (attempts_.back() == limit) ++round_;
    }
    dispatched_ += ids.size();
    return ids;
  }

  int record(const int* molIds, const int16_t* finishedOnIteration, int n) {
    const std::lock_guard<std::mutex> lock(mutex_);
    for (int i = 0; i < n; ++i) {
      if (molIds[i] < 0 || molIds[i] >= static_cast<int>(completed_.size())) return -1;
    }
    for (int i = 0; i < n; ++i) completed_[static_cast<size_t>(molIds[i])] += finishedOnIteration[i] == -1 ? 0 : 1;
    return 0;
  }

 private:
  std::mutex       mutex_;
  int              confs_;
  int              maxTries_;
  int              round_ = 1;
  uint64_t         dispatched_ = 0;
  std::vector<int> completed_;
  std::vector<int> attempts_;
};

// ---- control kernels (src/etkdg_kernels.cu:20-70) -------------------------------------------------------
__global__ void set_run_filter_kernel(const int n, uint8_t* __restrict__ active, const int16_t* __restrict__ finishedOn) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) active[i] = finishedOn[i] < 0;
}
__global__ void collect_failures_kernel(const int n, const uint8_t* __restrict__ failed, uint8_t* __restrict__ active,
                                        int16_t* __restrict__ failSum) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && active[i] && failed[i]) {
    active[i] = 0;
    failSum[i] += 1;
  }
}
__global__ void mark_finished_kernel(const int n, const int iteration, const uint8_t* __restrict__ active,
                                     int16_t* __restrict__ finishedOn, int* __restrict__ newlyFinished) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && finishedOn[i] == -1 && active[i]) {
    finishedOn[i] = static_cast<int16_t>(iteration);
    atomicAdd(newlyFinished, 1);
  }
}

// ---- start coordinates ----------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
// (rng - 0.5) * boxSize per coordinate of every ACTIVE system (src/etkdg_stage_coordgen.cu:100-121)
__global__ void random_coords_kernel(const int nSystems, const int32_t* __restrict__ atomStarts, const uint8_t* __restrict__ active,
                                     const uint64_t seed, const uint64_t attemptBase, const double boxSize,
                                     double* __restrict__ pos) {
  const int sys = blockIdx.x;
  if (sys >= nSystems || !active[sys]) return;
  const int c0 = atomStarts[sys] * 4, c1 = atomStarts[sys + 1] * 4;
  for (int c = c0 + threadIdx.x; c < c1; c += blockDim.x) {
    const uint64_t h = splitmix64(splitmix64(seed ^ ((attemptBase + sys) * 0x9e3779b97f4a7c15ull)) + static_cast<uint64_t>(c - c0));
    const double   u = static_cast<double>(h >> 11) * (1.0 / 9007199254740992.0);
    pos[c]           = (u - 0.5) * boxSize;
  }
}

// Surplus attempts leave before the second half of the pipeline.  The reference's rounds hand a molecule that misses one
// conformer another confs_per_mol attempts (Scheduler above), and nearly every attempt that survives the first minimisation
// and its checks is accepted in the end (failures per stage on the benchmark set: [0 23 458 0 0 0 1 0 5 0 0]) — so after stage
// 3 a molecule's attempts beyond what it still misses, plus ONE spare, would run the ETK minimisation (half of the pipeline's
// time) only to be dropped as extras when the batch is packed: 131 072 attempts for 97 951 conformers on 10 000 molecules x 10.
// Attempt s of a batch is kept when fewer than keep[s] active attempts of the same molecule precede it (attempts of a molecule
// are neighbours in a batch, in the scheduler's order, which is the order conformers are accepted in); the others are switched
// off without a failure being counted.  `before` is a snapshot of the active flags.
__global__ void prune_surplus_kernel(const int n, const int32_t* __restrict__ sysMol, const int32_t* __restrict__ keep,
                                     const uint8_t* __restrict__ before, uint8_t* __restrict__ active) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !before[i]) return;
  const int m    = sysMol[i];
  int       rank = 0;
  for (int j = i - 1; j >= 0 && sysMol[j] == m && rank < keep[i]; --j) rank += before[j];
  if (rank >= keep[i]) active[i] = 0;
}

// E / atom >= 0.05 after the first minimisation fails the attempt (etkdg_stage_distgeom_minimize.cu:36-51)
__global__ void energy_per_atom_check_kernel(const int n, const double* __restrict__ energies, const int32_t* __restrict__ atomStarts,
                                             uint8_t* __restrict__ failed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int na = atomStarts[i + 1] - atomStarts[i];
    if (na > 0 && energies[i] / na >= 0.05) failed[i] = 1;
  }
}
// planarity: improper-torsion energy > 0.7 * numImpropers fails (etkdg_stage_etk_minimization.cu:66-87)
__global__ void planar_check_kernel(const int n, const double* __restrict__ energies, const int32_t* __restrict__ sysMol,
                                    const int32_t* __restrict__ numImpropers, const uint8_t* __restrict__ active,
                                    uint8_t* __restrict__ failed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && active[i] && energies[i] > 0.7 * numImpropers[sysMol[i]]) failed[i] = 1;
}
// reference distances of the 1-2 / 1-3 restraints = current 3-D distances (etkdg_stage_etk_minimization.cu:32-64)
__global__ void reference_distance_kernel(const int nSystems, const int32_t* __restrict__ atomStarts,
                                          const int32_t* __restrict__ sysMol, const int32_t* __restrict__ termStarts,
                                          const int32_t* __restrict__ termIdx, const int32_t* __restrict__ refStarts,
                                          const double* __restrict__ pos, double* __restrict__ ref) {
  const int sys = blockIdx.x;
  if (sys >= nSystems) return;
  const int     m  = sysMol[sys];
  const int     t0 = termStarts[m], t1 = termStarts[m + 1];
  const double* p  = pos + static_cast<int64_t>(atomStarts[sys]) * 4;
  for (int t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
    const int    i = termIdx[2 * t], j = termIdx[2 * t + 1];
    const double dx = p[4 * i] - p[4 * j], dy = p[4 * i + 1] - p[4 * j + 1], dz = p[4 * i + 2] - p[4 * j + 2];
    ref[refStarts[sys] + t - t0] = sqrt(dx * dx + dy * dy + dz * dz);
  }
}

// ---- stereochemistry checks (src/etkdg_stage_stereochem_checks.cu:25-442) -------------------------------
struct P3 {
  double x, y, z;
};
__device__ __forceinline__ P3 sub(const P3 a, const P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ P3 crs(const P3 a, const P3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double dt(const P3 a, const P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// No guard for a zero vector, exactly like the reference's normalizeVector (src/forcefields/kernel_utils.cuh:140-145)
// and RDKit's Point3D::normalize: a three-coordinate centre repeats itself as its fourth neighbour, the fourth
// direction becomes NaN, every comparison with it is false and the volume test passes — returning the zero vector
// instead made all such centres FAIL (found by tests/test_stereo_checks_gpu.py against oracle/stereo.py).
__device__ __forceinline__ P3 unit(const P3 a) {
  const double l = sqrt(dt(a, a));
  return P3{a.x / l, a.y / l, a.z / l};
}
__device__ __forceinline__ P3 atom(const double* p, const int a) { return {p[4 * a], p[4 * a + 1], p[4 * a + 2]}; }

// is p0 on the same side of plane (v1, v2, v3) as v4, with both clear of the plane by `tol`? (:25-50)
__device__ __forceinline__ bool same_side(const double tol, const P3 v1, const P3 v2, const P3 v3, const P3 v4, const P3 p0) {
  const P3     n  = crs(sub(v2, v1), sub(v3, v1));
  const double d1 = dt(n, sub(v4, v1));
  const double d2 = dt(n, sub(p0, v1));
  if (fabs(d1) < tol || fabs(d2) < tol) return false;
  return !((d1 < 0.0) ^ (d2 < 0.0));
}

// centre 0 with neighbours 1-4 (a centre with three neighbours repeats itself as idx4)
__device__ bool tetrahedral_ok(const double* p, const int32_t* ix, const bool volumeTest, const bool fusedSmallRings, const double tol) {
  const P3 p0 = atom(p, ix[0]), p1 = atom(p, ix[1]), p2 = atom(p, ix[2]), p3 = atom(p, ix[3]), p4 = atom(p, ix[4]);
  if (volumeTest) {  // all four normalised triple products above MIN_TETRAHEDRAL_CHIRAL_VOL = 0.5 (x 0.25 in fused small rings)
    const P3     d1 = unit(sub(p0, p1)), d2 = unit(sub(p0, p2)), d3 = unit(sub(p0, p3)), d4 = unit(sub(p0, p4));
    const double lim = (fusedSmallRings ? 0.25 : 1.0) * 0.50;
    if (fabs(dt(crs(d1, d2), d3)) < lim) return false;
    if (fabs(dt(crs(d1, d2), d4)) < lim) return false;
    if (fabs(dt(crs(d1, d3), d4)) < lim) return false;
    if (fabs(dt(crs(d2, d3), d4)) < lim) return false;
  }
  if (ix[0] == ix[4]) return true;
  return same_side(tol, p1, p2, p3, p4, p0) && same_side(tol, p2, p3, p4, p1, p0) && same_side(tol, p3, p4, p1, p2, p0) &&
         same_side(tol, p4, p1, p2, p3, p0);
}

__global__ void stereo_check_kernel(const int nSystems, const int32_t* __restrict__ atomStarts, const int32_t* __restrict__ sysMol,
                                    const int32_t* __restrict__ checkStarts, const int32_t* __restrict__ checkKind,
                                    const int32_t* __restrict__ checkIdx, const double* __restrict__ checkPar, const int kind,
                                    const double* __restrict__ pos, const uint8_t* __restrict__ active, uint8_t* __restrict__ failed) {
  const int sys = blockIdx.x;
  if (sys >= nSystems || !active[sys]) return;
  const int     m = sysMol[sys];
  const double* p = pos + static_cast<int64_t>(atomStarts[sys]) * 4;
  for (int t = checkStarts[m] + threadIdx.x; t < checkStarts[m + 1]; t += blockDim.x) {
    if (checkKind[t] != kind) continue;
    const int32_t* ix   = checkIdx + 5 * t;
    const double   a    = checkPar[2 * t], b = checkPar[2 * t + 1];
    bool           fail = false;
    switch (kind) {
      case NVMK_CHECK_TETRAHEDRAL: fail = !tetrahedral_ok(p, ix, true, a != 0.0, 0.3); break;
      case NVMK_CHECK_CHIRAL_CENTER_VOLUME: fail = !tetrahedral_ok(p, ix, false, false, 0.1); break;
      case NVMK_CHECK_CHIRAL_VOLUME: {  // idx1..4 + [lb, ub] (:229-259)
        const P3     p4  = atom(p, ix[4]);
        const double vol = dt(sub(atom(p, ix[1]), p4), crs(sub(atom(p, ix[2]), p4), sub(atom(p, ix[3]), p4)));
        const bool   oppA = signbit(vol) != signbit(a), oppB = signbit(vol) != signbit(b);
        fail = (a > 0 && vol < a && (vol / a < 0.8 || oppA)) || (b < 0 && vol > b && (vol / b < 0.8 || oppB));
        break;
      }
      case NVMK_CHECK_CHIRAL_DISTANCE: {  // |d - bound| > 0.1 ub outside [lb, ub] (:261-301)
        const P3     d    = sub(atom(p, ix[0]), atom(p, ix[1]));
        const double dist = sqrt(dt(d, d));
        fail = (dist < a && fabs(dist - a) > 0.1 * b) || (dist > b && fabs(dist - b) > 0.1 * b);
        break;
      }
      case NVMK_CHECK_DOUBLE_BOND_STEREO: {  // dihedral 0-1-2-3 on the side given by sign = a (:303-377)
        const P3     p0 = atom(p, ix[0]), p1 = atom(p, ix[1]), p2 = atom(p, ix[2]), p3 = atom(p, ix[3]);
        const P3     r1 = sub(p2, p1), c1 = crs(sub(p0, p1), r1), c2 = crs(sub(p3, p2), r1);
        const double dot   = dt(c1, c2) / sqrt(dt(c1, c1) * dt(c2, c2));
        const double angle = dot <= -1.0 ? 3.14159265358979323846 : (dot >= 1.0 ? 0.0 : acos(dot));
        fail               = (angle - 1.57079632679489661923) * a < 0.0;
        break;
      }
      case NVMK_CHECK_DOUBLE_BOND_GEOMETRY: {  // 0-1-2 must not be linear (:379-442)
        const P3 u = unit(sub(atom(p, ix[1]), atom(p, ix[0]))), v = unit(sub(atom(p, ix[1]), atom(p, ix[2])));
        fail       = (dt(u, v) + 1.0) < 1.0e-3;
        break;
      }
      default: break;
    }
    if (fail) failed[sys] = 1;
  }
}

// accepted conformers -> 3-D output slots (src/conformer/etkdg_device_collect.cu packKernel4DTo3D)
__global__ void pack_kernel(const int nCopies, const int32_t* __restrict__ srcSystem, const int64_t* __restrict__ dstOffset,
                            const int32_t* __restrict__ atomStarts, const double* __restrict__ pos, double* __restrict__ out) {
  const int k = blockIdx.x;
  if (k >= nCopies) return;
  const int     sys = srcSystem[k];
  const int     a0 = atomStarts[sys], na = atomStarts[sys + 1] - a0;
  const double* p  = pos + static_cast<int64_t>(a0) * 4;
  double*       o  = out + dstOffset[k];
  for (int a = threadIdx.x; a < na; a += blockDim.x) {
    o[3 * a]     = p[4 * a];
    o[3 * a + 1] = p[4 * a + 1];
    o[3 * a + 2] = p[4 * a + 2];
  }
}

// The ETK terms only see x, y, z: the stage runs on a 3-D copy (a 4-D BFGS would carry 7/16 of its inverse Hessian as
// dead weight — the 4th coordinates have zero gradient and never couple) and the result is written back into the 4-D
// coordinates, 4th component untouched.
__global__ void copy_4d_to_3d_kernel(const int64_t nAtoms, const double* __restrict__ p4, double* __restrict__ p3) {
  const int64_t a = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (a >= nAtoms) return;
  p3[3 * a]     = p4[4 * a];
  p3[3 * a + 1] = p4[4 * a + 1];
  p3[3 * a + 2] = p4[4 * a + 2];
}
__global__ void copy_3d_to_4d_kernel(const int64_t nAtoms, const double* __restrict__ p3, double* __restrict__ p4) {
  const int64_t a = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (a >= nAtoms) return;
  p4[4 * a]     = p3[3 * a];
  p4[4 * a + 1] = p3[3 * a + 1];
  p4[4 * a + 2] = p3[3 * a + 2];
}

// Stage names as the reference's pipeline reports them (src/etkdg.cpp:339-380, etkdg_stage_*.h name()).
const char* const kStageNames[NVMK_ETKDG_N_STAGES] = {
  "Coordinate Generation", "First Minimization", "Tetrahedral Checks", "First Chirality Check", "Fourth Dimension Minimization",
  "ETK 3D Minimization", "Double bond geometry check", "Final Chirality Check", "Chirality Distance Matrix Check",
  "Final Chiral Center in Volume Check", "Double bond stereo check"};

// Per-stage wall clock of the last nvmk_etkdg_embed that ran with NVMK_ETKDG_TIMING=1 (the reference's debug-mode table,
// src/etkdg_impl.cpp:126-139,161-200): a stream synchronisation closes every stage, so the table costs throughput.  Two
// extra rows: the host work between batches (dispatch, uploads, record, pack) and the whole call.
constexpr int kTimingRows = NVMK_ETKDG_N_STAGES + 2;
struct StageTimings {
  double  total[kTimingRows] = {}, lo[kTimingRows] = {}, hi[kTimingRows] = {};
  int32_t calls[kTimingRows] = {};
  void    add(const int row, const double ms) {
    total[row] += ms;
    lo[row] = calls[row] == 0 ? ms : std::min(lo[row], ms);
    hi[row] = std::max(hi[row], ms);
    ++calls[row];
  }
};
std::mutex   g_timingMutex;
StageTimings g_lastTimings;
bool         g_haveTimings = false;

template <typename T> struct DevBuf {
  T*          p = nullptr;
  size_t      n = 0;
  hipError_t  ensure(size_t count) {
    if (count <= n) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};

inline unsigned blocks(int n, int b = 256) { return static_cast<unsigned>((n + b - 1) / b); }

}  // namespace etkdg
}  // namespace nvmk

using namespace nvmk;
using namespace nvmk::etkdg;

extern "C" {

// ---- scheduler handle (exposed so the reference's exact dispatch sequences can be tested) ----------------
void* nvmk_scheduler_create(int n_mols, int confs_per_mol, int max_iterations) {
  if (n_mols <= 0 || confs_per_mol <= 0 || max_iterations <= 0) {
    set_last_error("All parameters must be greater than 0.");  // src/etkdg_impl.cpp:277-279
    return nullptr;
  }
  return new Scheduler(n_mols, confs_per_mol, max_iterations);
}
void nvmk_scheduler_destroy(void* s) { delete static_cast<Scheduler*>(s); }
int  nvmk_scheduler_dispatch(void* s, int batch_size, int32_t* h_mol_ids_out, int* n_out) {
  NVMK_REQUIRE(s && n_out && (batch_size <= 0 || h_mol_ids_out), "scheduler dispatch: NULL argument");
  const std::vector<int> ids = static_cast<Scheduler*>(s)->dispatch(batch_size);
  for (size_t i = 0; i < ids.size(); ++i) h_mol_ids_out[i] = ids[i];
  *n_out = static_cast<int>(ids.size());
  return NVMK_OK;
}
int nvmk_scheduler_record(void* s, const int32_t* h_mol_ids, const int16_t* h_finished_on_iteration, int n) {
  NVMK_REQUIRE(s && (n == 0 || (h_mol_ids && h_finished_on_iteration)), "scheduler record: NULL argument");
  NVMK_REQUIRE(static_cast<Scheduler*>(s)->record(h_mol_ids, h_finished_on_iteration, n) == 0, "molId is out of range");
  return NVMK_OK;
}

int nvmk_etkdg_stereo_check(int kind, int n_systems, const int32_t* d_atom_starts, const int32_t* d_sys_mol,
                            const int32_t* d_check_starts, const int32_t* d_check_kind, const int32_t* d_check_idx,
                            const double* d_check_par, const double* d_pos, const uint8_t* d_active, uint8_t* d_failed,
                            void* stream_) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(kind >= NVMK_CHECK_TETRAHEDRAL && kind <= NVMK_CHECK_DOUBLE_BOND_GEOMETRY, "stereo check: unknown kind %d", kind);
  NVMK_REQUIRE(n_systems >= 0, "stereo check: negative system count");
  if (n_systems == 0) return NVMK_OK;
  NVMK_REQUIRE(d_atom_starts && d_sys_mol && d_check_starts && d_check_kind && d_check_idx && d_check_par && d_pos && d_failed,
               "stereo check: NULL buffer");
  hipStream_t   stream = as_stream(stream_);
  StreamScratch allActive;
  const uint8_t* active = d_active;
  if (active == nullptr) {
    NVMK_HIP_CHECK(allActive.alloc(static_cast<size_t>(n_systems), stream));
    NVMK_HIP_CHECK(hipMemsetAsync(allActive.ptr, 1, static_cast<size_t>(n_systems), stream));
    active = allActive.as<uint8_t>();
  }
  hipLaunchKernelGGL(stereo_check_kernel, dim3(n_systems), dim3(64), 0, stream, n_systems, d_atom_starts, d_sys_mol, d_check_starts,
                     d_check_kind, d_check_idx, d_check_par, kind, d_pos, active, d_failed);
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

// Start coordinates of stage 0 as a unit of its own (E3): system s of the batch is attempt attempt_base + s.
int nvmk_etkdg_random_coords(uint64_t seed, uint64_t attempt_base, int n_systems, const int32_t* d_atom_starts,
                             const uint8_t* d_active, double box_size, double* d_pos, void* stream_) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(n_systems >= 0, "random coords: negative system count");
  if (n_systems == 0) return NVMK_OK;
  NVMK_REQUIRE(d_atom_starts && d_pos, "random coords: NULL buffer");
  hipStream_t    stream = as_stream(stream_);
  StreamScratch  allActive;
  const uint8_t* active = d_active;
  if (active == nullptr) {
    NVMK_HIP_CHECK(allActive.alloc(static_cast<size_t>(n_systems), stream));
    NVMK_HIP_CHECK(hipMemsetAsync(allActive.ptr, 1, static_cast<size_t>(n_systems), stream));
    active = allActive.as<uint8_t>();
  }
  hipLaunchKernelGGL(random_coords_kernel, dim3(n_systems), dim3(64), 0, stream, n_systems, d_atom_starts, active, seed, attempt_base,
                     box_size, d_pos);
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

// The driver's bookkeeping (run filter -> per-stage failure collection -> finished marking) with PROGRAMMED stages: stage
// k of iteration i fails exactly the systems flagged in h_failed[(k * max_iterations + i) * n_systems + s].  The same three
// kernels nvmk_etkdg_embed runs between its real stages; this is how the reference's ProgrammableStep matrices
// (tests/test_etkdg.cu:41-341) are reproduced.
int nvmk_etkdg_driver_run(int n_systems, int n_stages, int max_iterations, const uint8_t* h_failed, int16_t* h_fail_counts,
                          int16_t* h_finished_on, int32_t* h_n_finished, int32_t* h_iterations, void* stream_) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(n_systems > 0, "driver: no conformers");    // ETKDGDriver throws on an empty context (test NoConformers)
  NVMK_REQUIRE(n_stages > 0, "driver: no stages");         // ... and on an empty stage list (test NoStages)
  NVMK_REQUIRE(max_iterations >= 0 && h_failed && h_fail_counts && h_finished_on && h_n_finished && h_iterations,
               "driver: NULL argument");
  hipStream_t      stream = as_stream(stream_);
  const size_t     n      = static_cast<size_t>(n_systems);
  DevBuf<uint8_t>  dActive, dFailed;
  DevBuf<int16_t>  dFinished, dFailSum;
  DevBuf<int>      dCount;
  NVMK_HIP_CHECK(dActive.ensure(n));
  NVMK_HIP_CHECK(dFailed.ensure(n));
  NVMK_HIP_CHECK(dFinished.ensure(n));
  NVMK_HIP_CHECK(dFailSum.ensure(n * static_cast<size_t>(n_stages)));
  NVMK_HIP_CHECK(dCount.ensure(1));
  NVMK_HIP_CHECK(hipMemsetAsync(dFinished.p, 0xff, n * 2, stream));
  NVMK_HIP_CHECK(hipMemsetAsync(dFailSum.p, 0, n * static_cast<size_t>(n_stages) * 2, stream));
  int finished = 0, iteration = 0;
  while (finished < n_systems && iteration < max_iterations) {  // ETKDGDriver::run (src/etkdg_impl.cpp:144-149)
    hipLaunchKernelGGL(set_run_filter_kernel, dim3(blocks(n_systems)), dim3(256), 0, stream, n_systems, dActive.p, dFinished.p);
    for (int k = 0; k < n_stages; ++k) {
      const uint8_t* prog = h_failed + (static_cast<size_t>(k) * max_iterations + iteration) * n;
      NVMK_HIP_CHECK(hipMemcpyAsync(dFailed.p, prog, n, hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(collect_failures_kernel, dim3(blocks(n_systems)), dim3(256), 0, stream, n_systems, dFailed.p, dActive.p,
                         dFailSum.p + static_cast<size_t>(k) * n);
    }
    NVMK_HIP_CHECK(hipMemsetAsync(dCount.p, 0, sizeof(int), stream));
    hipLaunchKernelGGL(mark_finished_kernel, dim3(blocks(n_systems)), dim3(256), 0, stream, n_systems, iteration, dActive.p, dFinished.p,
                       dCount.p);
    NVMK_LAUNCH_CHECK();
    int now = 0;
    NVMK_HIP_CHECK(hipMemcpyAsync(&now, dCount.p, sizeof(int), hipMemcpyDeviceToHost, stream));
    NVMK_HIP_CHECK(hipStreamSynchronize(stream));
    finished += now;
    ++iteration;
  }
  NVMK_HIP_CHECK(hipMemcpyAsync(h_fail_counts, dFailSum.p, n * static_cast<size_t>(n_stages) * 2, hipMemcpyDeviceToHost, stream));
  NVMK_HIP_CHECK(hipMemcpyAsync(h_finished_on, dFinished.p, n * 2, hipMemcpyDeviceToHost, stream));
  NVMK_HIP_CHECK(hipStreamSynchronize(stream));
  *h_n_finished = finished;
  *h_iterations = iteration;
  return NVMK_OK;
}

int nvmk_etkdg_embed(const nvmk_etkdg_molset* ms, const nvmk_etkdg_params* prm, double* d_coords, int32_t* h_conf_counts,
                     int32_t* h_stage_failures, void* stream_) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(ms && prm && h_conf_counts, "etkdg: NULL argument");
  NVMK_REQUIRE(ms->n_mols >= 0, "etkdg: negative molecule count");
  if (h_stage_failures) std::memset(h_stage_failures, 0, sizeof(int32_t) * NVMK_ETKDG_N_STAGES);
  if (ms->n_mols == 0) return NVMK_OK;
  NVMK_REQUIRE(ms->h_n_atoms && d_coords, "etkdg: NULL buffer");
  NVMK_REQUIRE(prm->confs_per_mol > 0 && prm->max_iterations > 0 && prm->batch_size > 0,
               "etkdg: confs_per_mol, max_iterations and batch_size must be > 0");
  hipStream_t stream = as_stream(stream_);
  const int   nMols  = ms->n_mols;
  const bool  useEtk = prm->use_exp_torsions != 0 || prm->use_basic_knowledge != 0;
  if (useEtk) NVMK_REQUIRE(ms->h_etk_d12_counts && ms->h_etk_d13_counts, "etkdg: ETK stage needs h_etk_d12/13_counts");
  if (prm->use_basic_knowledge) NVMK_REQUIRE(ms->num_impropers, "etkdg: basic-knowledge planarity check needs num_impropers");

  std::vector<int64_t> slotStart(static_cast<size_t>(nMols) + 1, 0);
  for (int m = 0; m < nMols; ++m) {
    NVMK_REQUIRE(ms->h_n_atoms[m] >= 0, "etkdg: negative atom count");
    slotStart[static_cast<size_t>(m) + 1] =
      slotStart[static_cast<size_t>(m)] + static_cast<int64_t>(ms->h_n_atoms[m]) * prm->confs_per_mol * 3;
    h_conf_counts[m] = 0;
  }

  // One batch at a time and the reference's rounds: measured against a demand-driven hand-out (a molecule is handed what it is
  // expected to still need) and against two to four concurrent batches, both slower at 10 000 molecules — the rounds' surplus
  // attempts (131 072 for 97 951 conformers) cost less than the small, latency-bound batches a frugal hand-out ends a run with
  // (profiles/r04_conformers/ab_etkdg_handout_and_workers_rejected.txt).
  Scheduler        sched(nMols, prm->confs_per_mol, prm->max_iterations);
  std::mutex       outMutex;    // h_conf_counts / output slots / h_stage_failures / timings
  const bool       timing = opt::get(opt::kEtkdgTiming).is("1");
  const bool       pruneSurplus = !opt::get(opt::kEtkdgPrune).is("0");  // NVMK_ETKDG_PRUNE=0: every attempt runs every stage
  StageTimings     timings;
  using Clock = std::chrono::steady_clock;
  auto ms_since = [](const Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); };
  const Clock::time_point tCall = Clock::now();
  std::atomic<int> firstError{NVMK_OK};

  // One worker = one stream running whole batches (dispatch -> 11 stages -> record -> pack) until the scheduler is
  // dry.  batches_per_gpu workers run concurrently (the reference's batchesPerGpu, src/etkdg.cpp:330-380): the long
  // tail of one batch (a handful of systems still iterating, the GPU almost idle) overlaps with the bulk of another.
  auto worker = [&](hipStream_t stream) -> int {
  DevBuf<int32_t>  dAtomStarts, dSysMol, dRef12Starts, dRef13Starts, dSrcSys, dKeep;
  DevBuf<int64_t>  dDstOff;
  DevBuf<double>   dPos, dPosMid, dPos3, dEnergies, dRef12, dRef13;
  DevBuf<uint8_t>  dActive, dFailed, dSub;
  DevBuf<int16_t>  dFinished, dStatuses, dFailSum;
  DevBuf<int>      dCount;
  NVMK_HIP_CHECK(dCount.ensure(1));

  for (;;) {
    if (firstError.load() != NVMK_OK) return NVMK_OK;
    const Clock::time_point tHost = Clock::now();
    double                  hostMs = 0.0;  // host work of this batch outside the stages
    NVMK_MARK("ETKDG batch");
    uint64_t               attemptBase = 0;
    std::vector<int> ids = sched.dispatch(prm->batch_size, &attemptBase);
    if (ids.empty()) break;
    // Largest systems first (stable): one workgroup minimises one system and the launch ends with the slowest one, so
    // the long jobs must not be the last to get a slot (cost per BFGS iteration grows with atoms^2).
    std::stable_sort(ids.begin(), ids.end(), [&](const int a, const int b) { return ms->h_n_atoms[a] > ms->h_n_atoms[b]; });
    const int nSys = static_cast<int>(ids.size());
    if (ms->build_handle != nullptr) {  // the tables of this batch's molecules (an asynchronous build fills the later ones meanwhile)
      int last = 0;
      for (const int m : ids) last = std::max(last, m);
      const int rcw = nvmk_etkdg_molset_wait(ms->build_handle, last + 1, stream);
      if (rcw != NVMK_OK) return rcw;
    }
    std::vector<int32_t> atomStarts(static_cast<size_t>(nSys) + 1, 0), r12(static_cast<size_t>(nSys) + 1, 0),
      r13(static_cast<size_t>(nSys) + 1, 0);
    for (int s = 0; s < nSys; ++s) {
      const int m = ids[static_cast<size_t>(s)];
      atomStarts[static_cast<size_t>(s) + 1] = atomStarts[static_cast<size_t>(s)] + ms->h_n_atoms[m];
      if (useEtk) {
        r12[static_cast<size_t>(s) + 1] = r12[static_cast<size_t>(s)] + ms->h_etk_d12_counts[m];
        r13[static_cast<size_t>(s) + 1] = r13[static_cast<size_t>(s)] + ms->h_etk_d13_counts[m];
      }
    }
    const int nAtoms = atomStarts.back();
    // attempts of a molecule that may go on past stage 3: what it still misses, and one spare (prune_surplus_kernel)
    std::vector<int32_t> keep(static_cast<size_t>(nSys), prm->confs_per_mol);
    if (pruneSurplus) {
      const std::lock_guard<std::mutex> lock(outMutex);
      for (int s = 0; s < nSys; ++s) keep[static_cast<size_t>(s)] = prm->confs_per_mol - h_conf_counts[ids[static_cast<size_t>(s)]] + 1;
      NVMK_HIP_CHECK(dKeep.ensure(static_cast<size_t>(nSys)));
      NVMK_HIP_CHECK(hipMemcpyAsync(dKeep.p, keep.data(), static_cast<size_t>(nSys) * 4, hipMemcpyHostToDevice, stream));
    }
    NVMK_HIP_CHECK(dAtomStarts.ensure(atomStarts.size()));
    NVMK_HIP_CHECK(dSysMol.ensure(static_cast<size_t>(nSys)));
    NVMK_HIP_CHECK(dPos.ensure(static_cast<size_t>(nAtoms) * 4));
    NVMK_HIP_CHECK(dPosMid.ensure(static_cast<size_t>(nAtoms) * 4));
    if (useEtk) NVMK_HIP_CHECK(dPos3.ensure(static_cast<size_t>(nAtoms) * 3));
    NVMK_HIP_CHECK(dEnergies.ensure(static_cast<size_t>(nSys)));
    NVMK_HIP_CHECK(dActive.ensure(static_cast<size_t>(nSys)));
    NVMK_HIP_CHECK(dFailed.ensure(static_cast<size_t>(nSys)));
    NVMK_HIP_CHECK(dSub.ensure(static_cast<size_t>(nSys)));
    NVMK_HIP_CHECK(dFinished.ensure(static_cast<size_t>(nSys)));
    NVMK_HIP_CHECK(dStatuses.ensure(static_cast<size_t>(nSys)));
    NVMK_HIP_CHECK(dFailSum.ensure(static_cast<size_t>(nSys) * NVMK_ETKDG_N_STAGES));
    NVMK_HIP_CHECK(hipMemcpyAsync(dAtomStarts.p, atomStarts.data(), atomStarts.size() * 4, hipMemcpyHostToDevice, stream));
    NVMK_HIP_CHECK(hipMemcpyAsync(dSysMol.p, ids.data(), static_cast<size_t>(nSys) * 4, hipMemcpyHostToDevice, stream));
    NVMK_HIP_CHECK(hipMemsetAsync(dFinished.p, 0xff, static_cast<size_t>(nSys) * 2, stream));  // -1
    NVMK_HIP_CHECK(hipMemsetAsync(dFailSum.p, 0, static_cast<size_t>(nSys) * NVMK_ETKDG_N_STAGES * 2, stream));
    if (useEtk) {
      NVMK_HIP_CHECK(dRef12Starts.ensure(r12.size()));
      NVMK_HIP_CHECK(dRef13Starts.ensure(r13.size()));
      NVMK_HIP_CHECK(dRef12.ensure(static_cast<size_t>(r12.back())));
      NVMK_HIP_CHECK(dRef13.ensure(static_cast<size_t>(r13.back())));
      NVMK_HIP_CHECK(hipMemcpyAsync(dRef12Starts.p, r12.data(), r12.size() * 4, hipMemcpyHostToDevice, stream));
      NVMK_HIP_CHECK(hipMemcpyAsync(dRef13Starts.p, r13.data(), r13.size() * 4, hipMemcpyHostToDevice, stream));
    }
    NVMK_HIP_CHECK(hipStreamSynchronize(stream));  // host vectors are reused below

    nvmk_ff_batch dg{};
    dg.kind = NVMK_FF_DG;
    dg.n_systems   = nSys;
    dg.atom_starts = dAtomStarts.p;
    dg.system_mol  = dSysMol.p;
    for (int g = 0; g < 3; ++g) dg.groups[g] = ms->dg[g];
    nvmk_ff_batch etk{};
    etk.kind = NVMK_FF_ETK;
    etk.n_systems   = nSys;
    etk.atom_starts = dAtomStarts.p;
    etk.system_mol  = dSysMol.p;
    for (int g = 0; g < 6; ++g) etk.groups[g] = ms->etk[g];

    int               stage = 0;
    Clock::time_point tStage;
    double            stageMs[NVMK_ETKDG_N_STAGES] = {};
    auto begin_stage = [&]() -> int {
      mark::push(kStageNames[stage]);
      if (timing) {
        if (stage == 0) {
          NVMK_HIP_CHECK(hipStreamSynchronize(stream));
          hostMs += ms_since(tHost);
        }
        tStage = Clock::now();
      }
      NVMK_HIP_CHECK(hipMemsetAsync(dFailed.p, 0, static_cast<size_t>(nSys), stream));
      return NVMK_OK;
    };
    auto end_stage = [&]() -> int {
      hipLaunchKernelGGL(collect_failures_kernel, dim3(blocks(nSys)), dim3(256), 0, stream, nSys, dFailed.p, dActive.p,
                         dFailSum.p + static_cast<size_t>(stage) * nSys);
      mark::pop();
      NVMK_LAUNCH_CHECK();
      if (timing) {
        NVMK_HIP_CHECK(hipStreamSynchronize(stream));
        stageMs[stage] = ms_since(tStage);
      }
      ++stage;
      return NVMK_OK;
    };
    auto check = [&](int kind, const double* pos4 = nullptr) -> int {
      if (ms->check_starts == nullptr) return NVMK_OK;
      hipLaunchKernelGGL(stereo_check_kernel, dim3(nSys), dim3(64), 0, stream, nSys, dAtomStarts.p, dSysMol.p, ms->check_starts,
                         ms->check_kind, ms->check_idx, ms->check_par, kind, pos4 != nullptr ? pos4 : dPos.p, dActive.p, dFailed.p);
      NVMK_LAUNCH_CHECK();
      return NVMK_OK;
    };
#define NVMK_TRY(expr)             \
  do {                             \
    const int rc_ = (expr);        \
    if (rc_ != NVMK_OK) return rc_; \
  } while (0)

    hipLaunchKernelGGL(set_run_filter_kernel, dim3(blocks(nSys)), dim3(256), 0, stream, nSys, dActive.p, dFinished.p);
    // stage 0: random start coordinates
    NVMK_TRY(begin_stage());
    hipLaunchKernelGGL(random_coords_kernel, dim3(nSys), dim3(64), 0, stream, nSys, dAtomStarts.p, dActive.p, prm->seed, attemptBase,
                       prm->box_size, dPos.p);
    NVMK_TRY(end_stage());
    // stage 1: first minimisation (chiral 1.0, 4th dim 0.1, 400 iterations) + energy check — and, in the same launch, stage
    // 4's fourth-dimension minimisation (chiral 0.2, 4th dim 1.0, 200 iterations) of every system right behind its first
    // one: stage 4 needs two iterations on average but a few systems take all 200, and as a launch of its own it ended every
    // batch with ~10 ms of an almost idle chip.  The coordinates in between stay in dPosMid for the checks of stages 1-3; a
    // system those checks fail has been minimised in vain, which changes nothing for the others (a system's minimisation
    // depends on its own coordinates only).  Systems 2 % beyond the energy limit of stage 1 skip the second minimisation.
    NVMK_TRY(begin_stage());
    {
      NVMK_HIP_CHECK(hipMemcpyAsync(dSub.p, dActive.p, static_cast<size_t>(nSys), hipMemcpyDeviceToDevice, stream));
      NVMK_HIP_CHECK(hipMemcpyAsync(dPosMid.p, dPos.p, static_cast<size_t>(nAtoms) * 4 * sizeof(double), hipMemcpyDeviceToDevice, stream));
      nvmk_bfgs_second_stage second{};
      second.w0                         = 0.2;
      second.w1                         = 1.0;
      second.max_iters                  = 200;
      second.restarts                   = 49;
      second.d_pos_between              = dPosMid.p;
      second.skip_above_energy_per_atom = 0.05 * 1.02;
      NVMK_TRY(nvmk_bfgs_minimize_two_stages(&dg, atomStarts.data(), 1.0, 0.1, 400, 49, &second, prm->force_tol, 1, dPos.p, dSub.p,
                                             dEnergies.p, dStatuses.p, nullptr, stream));
    }
    NVMK_TRY(nvmk_ff_energy(&dg, 1.0, 0.1, dPosMid.p, nullptr, dEnergies.p, stream));
    hipLaunchKernelGGL(energy_per_atom_check_kernel, dim3(blocks(nSys)), dim3(256), 0, stream, nSys, dEnergies.p, dAtomStarts.p,
                       dFailed.p);
    NVMK_TRY(end_stage());
    // stage 2: tetrahedral centres; stage 3: first chiral check (on the coordinates the first minimisation left)
    NVMK_TRY(begin_stage());
    NVMK_TRY(check(NVMK_CHECK_TETRAHEDRAL, dPosMid.p));
    NVMK_TRY(end_stage());
    NVMK_TRY(begin_stage());
    if (prm->enforce_chirality) NVMK_TRY(check(NVMK_CHECK_CHIRAL_VOLUME, dPosMid.p));
    NVMK_TRY(end_stage());
    // stage 4: fourth-dimension minimisation — ran in stage 1's launch
    NVMK_TRY(begin_stage());
    NVMK_TRY(end_stage());
    if (pruneSurplus) {  // between the halves of the pipeline: surplus attempts of a molecule leave (no failure counted)
      NVMK_HIP_CHECK(hipMemcpyAsync(dSub.p, dActive.p, static_cast<size_t>(nSys), hipMemcpyDeviceToDevice, stream));
      hipLaunchKernelGGL(prune_surplus_kernel, dim3(blocks(nSys)), dim3(256), 0, stream, nSys, dSysMol.p, dKeep.p, dSub.p, dActive.p);
      NVMK_LAUNCH_CHECK();
    }
    // stage 5: ETK minimisation (300 iterations) + planarity check
    NVMK_TRY(begin_stage());
    if (useEtk) {
      hipLaunchKernelGGL(reference_distance_kernel, dim3(nSys), dim3(64), 0, stream, nSys, dAtomStarts.p, dSysMol.p,
                         ms->etk[2].starts, ms->etk[2].idx, dRef12Starts.p, dPos.p, dRef12.p);
      hipLaunchKernelGGL(reference_distance_kernel, dim3(nSys), dim3(64), 0, stream, nSys, dAtomStarts.p, dSysMol.p,
                         ms->etk[3].starts, ms->etk[3].idx, dRef13Starts.p, dPos.p, dRef13.p);
      etk.etk_ref12_starts = dRef12Starts.p;
      etk.etk_ref12        = dRef12.p;
      etk.etk_ref13_starts = dRef13Starts.p;
      etk.etk_ref13        = dRef13.p;
      etk.group_mask       = prm->use_basic_knowledge ? 0x3fu : 0x3du;  // plain mode drops the improper terms (ETKTerm::PLAIN)
      const unsigned aBlocks = static_cast<unsigned>(blocks(nAtoms));
      hipLaunchKernelGGL(copy_4d_to_3d_kernel, dim3(aBlocks), dim3(256), 0, stream, static_cast<int64_t>(nAtoms), dPos.p, dPos3.p);
      {  // minimise the ACTIVE systems on the 3-D copy (one call, no repeat: etkdg_stage_etk_minimization.cu:204-266)
        NVMK_HIP_CHECK(hipMemcpyAsync(dSub.p, dActive.p, static_cast<size_t>(nSys), hipMemcpyDeviceToDevice, stream));
        NVMK_TRY(nvmk_bfgs_minimize(&etk, atomStarts.data(), 1.0, 1.0, 300, prm->force_tol, 1, dPos3.p, dSub.p, dEnergies.p,
                                    dStatuses.p, nullptr, stream));
      }
      if (prm->use_basic_knowledge) {
        nvmk_ff_batch planar = etk;
        planar.group_mask    = 0x2u;
        NVMK_TRY(nvmk_ff_energy(&planar, 1.0, 1.0, dPos3.p, nullptr, dEnergies.p, stream));
        hipLaunchKernelGGL(planar_check_kernel, dim3(blocks(nSys)), dim3(256), 0, stream, nSys, dEnergies.p, dSysMol.p,
                           ms->num_impropers, dActive.p, dFailed.p);
      }
      hipLaunchKernelGGL(copy_3d_to_4d_kernel, dim3(aBlocks), dim3(256), 0, stream, static_cast<int64_t>(nAtoms), dPos3.p, dPos.p);
    }
    NVMK_TRY(end_stage());
    // stages 6-10: final geometry / chirality checks
    NVMK_TRY(begin_stage());
    NVMK_TRY(check(NVMK_CHECK_DOUBLE_BOND_GEOMETRY));
    NVMK_TRY(end_stage());
    const int finals[4] = {NVMK_CHECK_CHIRAL_VOLUME, NVMK_CHECK_CHIRAL_DISTANCE, NVMK_CHECK_CHIRAL_CENTER_VOLUME,
                           NVMK_CHECK_DOUBLE_BOND_STEREO};
    for (int k = 0; k < 4; ++k) {
      NVMK_TRY(begin_stage());
      if (prm->enforce_chirality) NVMK_TRY(check(finals[k]));
      NVMK_TRY(end_stage());
    }
#undef NVMK_TRY
    const Clock::time_point tTail = Clock::now();
    // finished = still active after every stage (getFinishedKernels, iteration 0 of this batch)
    NVMK_HIP_CHECK(hipMemsetAsync(dCount.p, 0, sizeof(int), stream));
    hipLaunchKernelGGL(mark_finished_kernel, dim3(blocks(nSys)), dim3(256), 0, stream, nSys, 0, dActive.p, dFinished.p, dCount.p);
    NVMK_LAUNCH_CHECK();
    std::vector<int16_t> finished(static_cast<size_t>(nSys));
    std::vector<int16_t> failSum(static_cast<size_t>(nSys) * NVMK_ETKDG_N_STAGES);
    NVMK_HIP_CHECK(hipMemcpyAsync(finished.data(), dFinished.p, finished.size() * 2, hipMemcpyDeviceToHost, stream));
    NVMK_HIP_CHECK(hipMemcpyAsync(failSum.data(), dFailSum.p, failSum.size() * 2, hipMemcpyDeviceToHost, stream));
    NVMK_HIP_CHECK(hipStreamSynchronize(stream));
    sched.record(ids.data(), finished.data(), nSys);
    // copy accepted conformers into their output slots (extras beyond confs_per_mol are dropped)
    std::vector<int32_t> src;
    std::vector<int64_t> dst;
    {
      const std::lock_guard<std::mutex> lock(outMutex);
      if (h_stage_failures) {
        for (int st = 0; st < NVMK_ETKDG_N_STAGES; ++st) {
          for (int s = 0; s < nSys; ++s) h_stage_failures[st] += failSum[static_cast<size_t>(st) * nSys + s];
        }
      }
      for (int s = 0; s < nSys; ++s) {
        const int m = ids[static_cast<size_t>(s)];
        if (finished[static_cast<size_t>(s)] >= 0 && h_conf_counts[m] < prm->confs_per_mol) {
          src.push_back(s);
          dst.push_back(slotStart[static_cast<size_t>(m)] + static_cast<int64_t>(h_conf_counts[m]) * ms->h_n_atoms[m] * 3);
          ++h_conf_counts[m];
        }
      }
    }
    if (!src.empty()) {
      NVMK_HIP_CHECK(dSrcSys.ensure(src.size()));
      NVMK_HIP_CHECK(dDstOff.ensure(dst.size()));
      NVMK_HIP_CHECK(hipMemcpyAsync(dSrcSys.p, src.data(), src.size() * 4, hipMemcpyHostToDevice, stream));
      NVMK_HIP_CHECK(hipMemcpyAsync(dDstOff.p, dst.data(), dst.size() * 8, hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(pack_kernel, dim3(static_cast<unsigned>(src.size())), dim3(64), 0, stream, static_cast<int>(src.size()),
                         dSrcSys.p, dDstOff.p, dAtomStarts.p, dPos.p, d_coords);
      NVMK_LAUNCH_CHECK();
      NVMK_HIP_CHECK(hipStreamSynchronize(stream));
    }
    if (timing) {
      hostMs += ms_since(tTail);
      const std::lock_guard<std::mutex> lock(outMutex);
      for (int st = 0; st < NVMK_ETKDG_N_STAGES; ++st) timings.add(st, stageMs[st]);
      timings.add(NVMK_ETKDG_N_STAGES, hostMs);
    }
  }
  return NVMK_OK;
  };  // worker
  auto publish = [&]() {
    if (!timing) return;
    timings.add(NVMK_ETKDG_N_STAGES + 1, ms_since(tCall));
    const std::lock_guard<std::mutex> lock(g_timingMutex);
    g_lastTimings = timings;
    g_haveTimings = true;
  };

  const int64_t totalAttempts = static_cast<int64_t>(nMols) * prm->confs_per_mol;
  int           nWorkers      = prm->batches_per_gpu > 1 ? prm->batches_per_gpu : 1;
  if (totalAttempts <= prm->batch_size) nWorkers = 1;
  if (nWorkers == 1) {
    const int rc1 = worker(stream);
    publish();
    return rc1;
  }

  int device = 0;
  NVMK_HIP_CHECK(hipGetDevice(&device));
  NVMK_HIP_CHECK(hipStreamSynchronize(stream));  // inputs enqueued on the caller's stream are complete
  std::vector<hipStream_t> streams(static_cast<size_t>(nWorkers), nullptr);
  for (auto& st : streams) NVMK_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  std::vector<std::string> errors(static_cast<size_t>(nWorkers));
  auto run = [&](const int w) {
    if (hipSetDevice(device) != hipSuccess) {
      firstError.store(NVMK_ERR_HIP);
      errors[static_cast<size_t>(w)] = "hipSetDevice failed in an ETKDG worker";
      return;
    }
    const int rc = worker(streams[static_cast<size_t>(w)]);
    if (rc != NVMK_OK) {
      int expected = NVMK_OK;
      firstError.compare_exchange_strong(expected, rc);
      errors[static_cast<size_t>(w)] = nvmk_last_error();  // thread-local message of this worker
    }
  };
  std::vector<std::thread> threads;
  for (int w = 1; w < nWorkers; ++w) threads.emplace_back(run, w);
  run(0);
  for (auto& t : threads) t.join();
  for (auto& st : streams) (void)hipStreamDestroy(st);  // every batch ended with a stream synchronisation
  if (firstError.load() != NVMK_OK) {
    for (const auto& e : errors) {
      if (!e.empty()) {
        ::nvmk::set_last_error("%s", e.c_str());
        break;
      }
    }
    return firstError.load();
  }
  publish();
  return NVMK_OK;
}

int nvmk_etkdg_stage_timings(double* total_ms, double* min_ms, double* max_ms, int32_t* calls, int n_rows, const char** names) {
  NVMK_REQUIRE(n_rows >= 0 && (n_rows == 0 || (total_ms && min_ms && max_ms && calls)), "stage timings: NULL buffer");
  const std::lock_guard<std::mutex> lock(g_timingMutex);
  NVMK_REQUIRE(g_haveTimings, "stage timings: no nvmk_etkdg_embed call has run with NVMK_ETKDG_TIMING=1");
  static const char* const kExtra[2] = {"host work between the stages (dispatch, uploads, record, pack)", "whole call"};
  for (int r = 0; r < std::min(n_rows, kTimingRows); ++r) {
    total_ms[r] = g_lastTimings.total[r];
    min_ms[r]   = g_lastTimings.lo[r];
    max_ms[r]   = g_lastTimings.hi[r];
    calls[r]    = g_lastTimings.calls[r];
    if (names) names[r] = r < NVMK_ETKDG_N_STAGES ? kStageNames[r] : kExtra[r - NVMK_ETKDG_N_STAGES];
  }
  return NVMK_OK;
}

}  // extern "C"
