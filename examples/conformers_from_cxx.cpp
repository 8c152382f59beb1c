// Host arrays -> conformers -> minimised conformers, from C++ with no Python anywhere: the call sequence the reference's
// src/etkdg.cpp (embedMolecules: prepareEmbedderArgs per molecule on OpenMP threads, then batches on the GPU, :175-240) and
// src/minimizer/bfgs_mmff.cpp (flatten once per molecule, add every conformer to a batch, minimise, :139-328) would make
// against this library:
//   1. per-molecule term arrays as a flattener leaves them (here: chain molecules whose distance bounds come from a hidden
//      geometry, so embeddings exist; bonded + van der Waals MMFF terms with that geometry's rest lengths),
//   2. nvmk_etkdg_molset_build      -> resident, kernel-ordered distance-geometry tables (host threads, pinned staging, async upload),
//   3. nvmk_ff_tables_build (MMFF)  -> the force-field tables, assembled on a second host thread WHILE step 4 runs,
//   4. nvmk_etkdg_embed             -> conformers in device memory,
//   5. nvmk_bfgs_minimize           -> every conformer minimised where it lies; conformers share their molecule's tables (system_mol).
// Checked on the host: every conformer's distance-violation energy (recomputed here) is below the pipeline's acceptance limit of
// 0.05 per atom, and the force-field energy the minimiser reports
// is lower than the energy of the embedded geometry (nvmk_ff_energy) for every conformer.
//   g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/conformers_from_cxx.cpp -Lnvmolkit_amd/lib \
//       -lnvmolkit_amd -L/opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,$PWD/nvmolkit_amd/lib -Wl,-rpath,/opt/rocm/lib -o conformers
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <future>
#include <random>
#include <vector>

extern "C" {
#include "nvmolkit_amd.h"
}

#define HIP_OK(call)                                                  \
  do {                                                                \
    const hipError_t e_ = (call);                                     \
    if (e_ != hipSuccess) {                                           \
      std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); \
      return 2;                                                       \
    }                                                                 \
  } while (0)
#define NVMK_DO(call)                                              \
  do {                                                             \
    if ((call) != NVMK_OK) {                                       \
      std::fprintf(stderr, "%s: %s\n", #call, nvmk_last_error()); \
      return 1;                                                    \
    }                                                              \
  } while (0)

namespace {

// What a flattener produces for one molecule; the descriptors below point into these vectors.
struct Molecule {
  int                  nAtoms = 0;
  std::vector<double>  ref;                       // hidden geometry, 3 per atom
  std::vector<int32_t> pairIdx;                   // all pairs (i < j), i-major — the order every natural builder emits
  std::vector<double>  pairBounds;                // (lb^2, ub^2, weight) per pair
  std::vector<int32_t> fourthDim;                 // every atom
  std::vector<int32_t> bondIdx, vdwIdx;           // MMFF: bonds (i, i + 1), van der Waals pairs (separation >= 3)
  std::vector<double>  bondPar, vdwPar;           // (r0, kb), (R*, eps)
};

Molecule make_chain(std::mt19937& rng, const int n) {
  std::normal_distribution<double> gauss;
  Molecule                         m;
  m.nAtoms = n;
  m.ref.assign(static_cast<size_t>(n) * 3, 0.0);
  for (int a = 1; a < n; ++a) {  // self-avoiding-ish random chain with 1.5 A steps
    double cand[3] = {};
    for (int attempt = 0; attempt < 50; ++attempt) {
      double s[3] = {gauss(rng), gauss(rng), gauss(rng)};
      const double len = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
      for (int k = 0; k < 3; ++k) cand[k] = m.ref[3 * (a - 1) + k] + 1.5 * s[k] / len;
      double closest = 1e9;
      for (int b = 0; b + 1 < a; ++b) {
        double d2 = 0;
        for (int k = 0; k < 3; ++k) d2 += (m.ref[3 * b + k] - cand[k]) * (m.ref[3 * b + k] - cand[k]);
        closest = std::min(closest, std::sqrt(d2));
      }
      if (a < 2 || closest > 1.1) break;
    }
    for (int k = 0; k < 3; ++k) m.ref[3 * a + k] = cand[k];
  }
  for (int i = 0; i < n; ++i) {
    m.fourthDim.push_back(i);
    for (int j = i + 1; j < n; ++j) {
      double d2 = 0;
      for (int k = 0; k < 3; ++k) d2 += (m.ref[3 * i + k] - m.ref[3 * j + k]) * (m.ref[3 * i + k] - m.ref[3 * j + k]);
      const double d = std::sqrt(d2), sep = j - i, tol = std::min(0.02 * sep * sep, 1.0);
      const double lb = std::max(d - tol, 0.5), ub = d + tol;
      m.pairIdx.insert(m.pairIdx.end(), {i, j});
      m.pairBounds.insert(m.pairBounds.end(), {lb * lb, ub * ub, 1.0});
      if (sep == 1) {
        m.bondIdx.insert(m.bondIdx.end(), {i, j});
        m.bondPar.insert(m.bondPar.end(), {d, 4.5});
      } else if (sep >= 3) {
        m.vdwIdx.insert(m.vdwIdx.end(), {i, j});
        m.vdwPar.insert(m.vdwPar.end(), {3.6, 0.05});
      }
    }
  }
  return m;
}

nvmk_host_terms terms(const std::vector<int32_t>& idx, const std::vector<double>& par, const int nIdx) {
  return {static_cast<int32_t>(idx.size() / static_cast<size_t>(nIdx)), 4, idx.data(), par.data()};
}

}  // namespace

int main() {
  const int    nMols = 64, confs = 4;
  std::mt19937 rng(11);
  std::vector<Molecule> mols;
  for (int m = 0; m < nMols; ++m) mols.push_back(make_chain(rng, 8 + static_cast<int>(rng() % 25)));

  // descriptors: pointers into the molecules' own arrays, nothing is copied here
  std::vector<nvmk_flat_molecule> flat(nMols);
  std::vector<nvmk_host_terms>    mmff(static_cast<size_t>(nMols) * 7);
  const std::vector<int32_t>      noIdx;
  const std::vector<double>       noPar;
  for (int m = 0; m < nMols; ++m) {
    const Molecule& x = mols[m];
    flat[m]           = nvmk_flat_molecule{};
    flat[m].n_atoms   = x.nAtoms;
    flat[m].dg[0]     = terms(x.pairIdx, x.pairBounds, 2);
    flat[m].dg[1]     = terms(noIdx, noPar, 4);          // no chiral centres
    flat[m].dg[2]     = terms(x.fourthDim, noPar, 1);
    for (int g = 0; g < 7; ++g) mmff[static_cast<size_t>(m) * 7 + g] = terms(noIdx, noPar, 1);
    mmff[static_cast<size_t>(m) * 7 + 0] = terms(x.bondIdx, x.bondPar, 2);
    mmff[static_cast<size_t>(m) * 7 + 5] = terms(x.vdwIdx, x.vdwPar, 2);
  }

  hipStream_t stream = nullptr, side = nullptr;
  HIP_OK(hipStreamCreate(&stream));
  HIP_OK(hipStreamCreate(&side));

  // 2. the embedding tables; 3. the force-field tables on a second host thread and stream while the embedding runs
  void* molsetHandle = nullptr;
  // (NVMK_BUILD_ASYNC: the call returns once the tables are planned; nvmk_etkdg_embed meets their rows batch by batch, so the
  // tables of batch k + 1 are assembled while batch k is on the GPU — `flat` and the arrays behind it stay alive until then)
  NVMK_DO(nvmk_etkdg_molset_build(flat.data(), nMols, /*n_threads=*/0, NVMK_BUILD_ASYNC, stream, &molsetHandle));
  nvmk_etkdg_molset molset;
  NVMK_DO(nvmk_etkdg_molset_view(molsetHandle, &molset));
  int device = 0;
  HIP_OK(hipGetDevice(&device));
  void*            tablesHandle = nullptr;
  std::future<int> pending      = std::async(std::launch::async, [&] {
    if (hipSetDevice(device) != hipSuccess) return NVMK_ERR_HIP;
    return nvmk_ff_tables_build(NVMK_FF_MMFF, mmff.data(), nMols, 7, 0, 0, side, &tablesHandle);
  });

  // 4. ETKDG without the ETK stage (plain distance geometry), conformer c of molecule m at 3 * (confs * atomsBefore[m] + c * n_atoms[m])
  std::vector<int64_t> atomsBefore(nMols + 1, 0);
  for (int m = 0; m < nMols; ++m) atomsBefore[m + 1] = atomsBefore[m] + mols[m].nAtoms;
  double* dCoords = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&dCoords), static_cast<size_t>(atomsBefore[nMols]) * confs * 3 * sizeof(double)));
  nvmk_etkdg_params prm{};
  prm.confs_per_mol  = confs;
  prm.max_iterations = 10;
  prm.batch_size     = 16384;
  prm.box_size       = 10.0;
  prm.force_tol      = 1e-3;
  prm.seed           = 42;
  std::vector<int32_t> counts(nMols), stageFailures(NVMK_ETKDG_N_STAGES);
  NVMK_DO(nvmk_etkdg_embed(&molset, &prm, dCoords, counts.data(), stageFailures.data(), stream));
  std::vector<double> coords(static_cast<size_t>(atomsBefore[nMols]) * confs * 3);
  HIP_OK(hipMemcpyAsync(coords.data(), dCoords, coords.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));

  int nConformers = 0, boundViolations = 0;
  for (int m = 0; m < nMols; ++m) {
    nConformers += counts[m];
    const Molecule& x = mols[m];
    for (int c = 0; c < counts[m]; ++c) {
      const double* p = coords.data() + 3 * (confs * atomsBefore[m] + static_cast<int64_t>(c) * x.nAtoms);
      double        energy = 0.0;  // the distance-violation term of src/forcefields/dist_geom_kernels_device.cuh:37-80
      for (size_t t = 0; t < x.pairIdx.size() / 2; ++t) {
        const int i = x.pairIdx[2 * t], j = x.pairIdx[2 * t + 1];
        double    d2 = 0;
        for (int k = 0; k < 3; ++k) d2 += (p[3 * i + k] - p[3 * j + k]) * (p[3 * i + k] - p[3 * j + k]);
        const double lb2 = x.pairBounds[3 * t], ub2 = x.pairBounds[3 * t + 1];
        const double val = d2 > ub2 ? d2 / ub2 - 1.0 : (d2 < lb2 ? 2.0 * lb2 / (lb2 + d2) - 1.0 : 0.0);
        energy += val * val;
      }
      if (!(energy / x.nAtoms < 0.05)) ++boundViolations;
    }
  }
  std::printf("ETKDG: %d conformers of %d molecules x %d, %d with a distance-violation energy above 0.05 per atom\n", nConformers, nMols, confs,
              boundViolations);

  // 5. one batch of all conformers: positions packed conformer after conformer, tables shared through system_mol
  if (pending.get() != NVMK_OK) {
    std::fprintf(stderr, "nvmk_ff_tables_build: failed on the side thread\n");
    return 1;
  }
  NVMK_DO(nvmk_ff_tables_wait(tablesHandle, stream));  // built under `side`: the stream the minimisation runs on waits for the uploads
  std::vector<int32_t> atomStarts{0}, systemMol;
  std::vector<double>  pos;
  for (int m = 0; m < nMols; ++m) {
    for (int c = 0; c < counts[m]; ++c) {
      const double* p = coords.data() + 3 * (confs * atomsBefore[m] + static_cast<int64_t>(c) * mols[m].nAtoms);
      pos.insert(pos.end(), p, p + 3 * mols[m].nAtoms);
      atomStarts.push_back(atomStarts.back() + mols[m].nAtoms);
      systemMol.push_back(m);
    }
  }
  const int nSys = static_cast<int>(systemMol.size());
  int32_t * dAtomStarts = nullptr, *dSystemMol = nullptr, *dIters = nullptr;
  double *  dPos = nullptr, *dBefore = nullptr, *dAfter = nullptr;
  int16_t*  dStatus = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&dAtomStarts), atomStarts.size() * 4));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&dSystemMol), std::max<size_t>(systemMol.size(), 1) * 4));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&dPos), std::max<size_t>(pos.size(), 1) * 8));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&dBefore), std::max(nSys, 1) * 8));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&dAfter), std::max(nSys, 1) * 8));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&dStatus), std::max(nSys, 1) * 2));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&dIters), std::max(nSys, 1) * 4));
  HIP_OK(hipMemcpyAsync(dAtomStarts, atomStarts.data(), atomStarts.size() * 4, hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(dSystemMol, systemMol.data(), systemMol.size() * 4, hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(dPos, pos.data(), pos.size() * 8, hipMemcpyHostToDevice, stream));
  nvmk_ff_batch batch{};
  batch.kind        = NVMK_FF_MMFF;
  batch.n_systems   = nSys;
  batch.atom_starts = dAtomStarts;
  batch.system_mol  = dSystemMol;
  NVMK_DO(nvmk_ff_tables_view(tablesHandle, batch.groups, nullptr));
  NVMK_DO(nvmk_ff_energy(&batch, 1.0, 1.0, dPos, nullptr, dBefore, stream));
  NVMK_DO(nvmk_bfgs_minimize(&batch, atomStarts.data(), 1.0, 1.0, /*max_iters=*/200, /*grad_tol=*/1e-4, /*scale_grads=*/1, dPos, nullptr, dAfter,
                             dStatus, dIters, stream));
  std::vector<double>  before(nSys), after(nSys);
  std::vector<int16_t> status(nSys);
  HIP_OK(hipMemcpyAsync(before.data(), dBefore, before.size() * 8, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(after.data(), dAfter, after.size() * 8, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(status.data(), dStatus, status.size() * 2, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  int notLower = 0, converged = 0;
  for (int s = 0; s < nSys; ++s) {
    if (!(after[s] <= before[s]) || !std::isfinite(after[s])) ++notLower;
    if (status[s] == 0) ++converged;
  }
  std::printf("MMFF: %d conformers minimised, %d converged, %d whose energy did not go down\n", nSys, converged, notLower);

  NVMK_DO(nvmk_ff_tables_free(tablesHandle));
  NVMK_DO(nvmk_etkdg_molset_free(molsetHandle));
  for (void* p : {static_cast<void*>(dCoords), static_cast<void*>(dAtomStarts), static_cast<void*>(dSystemMol), static_cast<void*>(dPos),
                  static_cast<void*>(dBefore), static_cast<void*>(dAfter), static_cast<void*>(dStatus), static_cast<void*>(dIters)})
    (void)hipFree(p);
  (void)hipStreamDestroy(stream);
  (void)hipStreamDestroy(side);
  const bool ok = nConformers >= nMols * confs * 9 / 10 && boundViolations == 0 && notLower == 0 && nSys == nConformers;
  std::printf("%s\n", ok ? "conformers from C++: OK" : "conformers from C++: FAILED");
  return ok ? 0 : 1;
}
