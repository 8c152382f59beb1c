// Microbenchmark of the inverse-Hessian pass of the fused BFGS kernels (nvmolkit_amd/csrc/hess_pass.h) on its own:
// `systems` workgroups, each owning a packed n x n inverse Hessian (rows resident in LDS as far as the LDS budget goes,
// the rest in HBM, exactly like bfgs_kernel), run `iters` passes with a pending rank-2 update.  Prints the time per pass
// (whole launch / iters, and the mean of the workgroups' own clocks) and the HBM bytes the passes requested.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/ubench_hess.hip -o tools/ubench_hess
//     [-DUBENCH_THREADS=64|128|256]  workgroup size (one, two or four waves per system; default 256)
//     (with tools/experiments/hess_exact_rows.patch applied: -DNVMK_HESS_EXACT=0|1, -DNVMK_HESS_AUX_LOAD=2 -DNVMK_HESS_AUX_STORE=2 —
//     HBM rows through per-row buffer descriptors, measured and rejected in round 4, profiles/r04_conformers/)
//   tools/ubench_hess [n=192] [systems=4096] [ldsKB=79] [occ=2] [iters=50]
// (ldsKB may be fractional: 19.5 = the share of a one-wave workgroup when eight of them sit on a CU)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef UBENCH_THREADS
#define UBENCH_THREADS 256
#endif
#define NVMK_BFGS_NS ub
#define NVMK_BFGS_THREADS UBENCH_THREADS
#include "../nvmolkit_amd/csrc/hess_pass.h"

using namespace nvmk::minim::ub;
#ifndef NVMK_HESS_EXACT
#define NVMK_HESS_EXACT 0
#define NVMK_HESS_AUX_LOAD 0
#define NVMK_HESS_AUX_STORE 0
#endif

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));             \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

template <int OCC>
__global__ __launch_bounds__(NT, OCC) void pass_kernel(double* __restrict__ hessians, const int64_t* __restrict__ starts, const int n,
                                                       const int ldsDoubles, const int iters, double* __restrict__ checksum,
                                                       long long* __restrict__ ticks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  double*   xi   = reinterpret_cast<double*>(smem);
  double*   hdg  = xi + n;
  double*   uu   = hdg + n;
  double*   g    = uu + n;
  double*   tvec = g + n;
  double*   diag = xi + 9 * n;  // same offsets as bfgs_kernel: 9 vectors, the diagonal, then the partial sums
  double*   part = diag + n;
  double*   red  = part + (1 + NW) * n;
  double*   Hl   = red + kRedDoubles;
  double*   H    = hessians + starts[blockIdx.x];
  const int Rl   = resident_rows(n, lds_hessian_doubles(ldsDoubles, n));
  for (int i = tid; i < n; i += NT) {
    xi[i]  = 1.0e-3 * ((i * 37 + blockIdx.x) % 101 - 50);
    hdg[i] = 1.0e-3 * ((i * 53 + 7) % 89 - 44);
    uu[i]  = 1.0e-3 * ((i * 11 + 3) % 97 - 48);
    g[i]   = 1.0e-2 * ((i * 29 + 5) % 83 - 41);
  }
  {
    const int64_t nl = hess_row_offset(Rl), total = hess_row_offset(n);
    for (int64_t i = tid; i < nl; i += NT) Hl[i] = 0.0;
    for (int64_t i = tid; i < total - nl; i += NT) H[i] = 0.0;
    for (int r = tid; r < n; r += NT) diag[r] = 1.0;
  }
  __syncthreads();
  const long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    hess_pass<(OCC < 3)>(diag, Hl, H, Rl, n, true, 1.0e-6, 2.0e-6, 0.5e-3, xi, hdg, uu, g, part);
    hess_finish(n, part, tvec);
    __syncthreads();
    for (int i = tid; i < n; i += NT) g[i] = 0.5 * g[i] + 1.0e-3 * tvec[i];  // the next pass depends on this one
    __syncthreads();
  }
  const long long t1 = wall_clock64();
  if (tid == 0) {
    ticks[blockIdx.x] = t1 - t0;
    double s          = 0.0;
    for (int i = 0; i < n; ++i) s += tvec[i];
    checksum[blockIdx.x] = s;
  }
}

int main(int argc, char** argv) {
  const int n       = argc > 1 ? std::atoi(argv[1]) : 192;
  const int systems = argc > 2 ? std::atoi(argv[2]) : 4096;
  const double ldsKB = argc > 3 ? std::atof(argv[3]) : 79;
  const int occ     = argc > 4 ? std::atoi(argv[4]) : 2;
  const int iters   = argc > 5 ? std::atoi(argv[5]) : 50;
  const size_t vecBytes = static_cast<size_t>(lds_vector_doubles(n)) * 8;
  size_t       shmem    = std::max<size_t>(vecBytes, static_cast<size_t>(ldsKB * 1024) & ~size_t{15});
  shmem                 = std::min(shmem, vecBytes + static_cast<size_t>(hess_row_offset(n)) * 8);
  const int ldsDoubles  = static_cast<int>(shmem / 8);
  const int rl          = resident_rows(n, lds_hessian_doubles(ldsDoubles, n));
  const int64_t perSys  = hess_row_offset(n) - hess_row_offset(rl);
  std::vector<int64_t> starts(static_cast<size_t>(systems) + 1);
  for (int s = 0; s <= systems; ++s) starts[static_cast<size_t>(s)] = perSys * s;
  double *   dH = nullptr, *dSum = nullptr;
  int64_t*   dStarts = nullptr;
  long long* dTicks  = nullptr;
  CHECK(hipMalloc(&dH, (static_cast<size_t>(perSys) * systems + kHessTailPadDoubles) * 8));
  CHECK(hipMalloc(&dSum, systems * sizeof(double)));
  CHECK(hipMalloc(&dStarts, starts.size() * sizeof(int64_t)));
  CHECK(hipMalloc(&dTicks, systems * sizeof(long long)));
  CHECK(hipMemcpy(dStarts, starts.data(), starts.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  auto launch = [&](int its) -> hipError_t {
    if (occ >= 3) {
      if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pass_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shmem));
      hipLaunchKernelGGL(pass_kernel<3>, dim3(systems), dim3(NT), shmem, nullptr, dH, dStarts, n, ldsDoubles, its, dSum, dTicks);
    } else {
      if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pass_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shmem));
      hipLaunchKernelGGL(pass_kernel<2>, dim3(systems), dim3(NT), shmem, nullptr, dH, dStarts, n, ldsDoubles, its, dSum, dTicks);
    }
    return hipGetLastError();
  };
  CHECK(launch(2));
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  CHECK(launch(iters));
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> ticks(static_cast<size_t>(systems));
  std::vector<double>    sums(static_cast<size_t>(systems));
  CHECK(hipMemcpy(ticks.data(), dTicks, ticks.size() * sizeof(long long), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(sums.data(), dSum, sums.size() * sizeof(double), hipMemcpyDeviceToHost));
  double meanTicks = 0.0;
  for (long long t : ticks) meanTicks += static_cast<double>(t);
  meanTicks /= systems;
  const double hbmBytes = static_cast<double>(perSys) * 16.0 * systems * iters;
  std::printf("{\"threads\": %d, \"exact\": %d, \"aux\": [%d, %d], \"n\": %d, \"systems\": %d, \"lds_bytes\": %zu, \"occ\": %d, \"resident_rows\": %d, \"us_per_pass_per_workgroup\": %.2f, "
              "\"launch_ms\": %.3f, \"passes_per_us_whole_gpu\": %.2f, \"hbm_GBps\": %.0f, \"checksum\": %.12g}\n",
              NT, NVMK_HESS_EXACT, NVMK_HESS_AUX_LOAD, NVMK_HESS_AUX_STORE, n, systems, shmem, occ, rl, meanTicks * 0.01 / iters, ms, static_cast<double>(systems) * iters / (ms * 1e3),
              hbmBytes / (ms * 1e-3) / 1e9, sums[0]);
  return 0;
}
