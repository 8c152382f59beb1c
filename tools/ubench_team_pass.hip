// Microbenchmark of the cooperative class's inverse-Hessian pass (hess_pass_rows, nvmolkit_amd/csrc/hess_pass.h) on its own: the
// chip is filled with workgroups, `width` of them share one packed n x n triangle the way a team's ranks do (contiguous row
// blocks of equal cost), every workgroup runs `iters` passes over its block with a pending rank-2 update.  No barriers between
// the ranks, no force field, no vector work: what is measured is how fast this access pattern streams.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/ubench_team_pass.hip -o /tmp/ubench_team_pass [-DUBENCH_THREADS=512|256]
//   /tmp/ubench_team_pass [n=4252] [width=32] [workgroups=256] [iters=20] [rowsCap=2048]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef UBENCH_THREADS
#define UBENCH_THREADS 512
#endif
#define NVMK_BFGS_NS ub
#define NVMK_BFGS_THREADS UBENCH_THREADS
#include "../nvmolkit_amd/csrc/hess_pass.h"

using namespace nvmk::minim::ub;

#define CHECK(x)                                                          \
  do {                                                                    \
    hipError_t e_ = (x);                                                  \
    if (e_ != hipSuccess) {                                               \
      std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); \
      return 1;                                                           \
    }                                                                     \
  } while (0)

static int row_end(const int n, const int size, const int q) {  // bfgs_device.inc: team_row_end
  if (q >= size - 1) return n;
  const double total  = 0.5 * double(n) * double(n) + 64.0 * double(n);
  const double target = total * double(q + 1) / double(size);
  const int    r      = static_cast<int>(std::sqrt(64.0 * 64.0 + 2.0 * target) - 64.0);
  return std::min(n, (r + 3) & ~3);
}

__global__ __launch_bounds__(NT, 2) void pass_kernel(double* __restrict__ hessians, const int64_t slot, double* __restrict__ vecs,
                                                     const int n, const int width, const int* __restrict__ ends, const int iters,
                                                     const int rowsCap, double* __restrict__ checksum) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid  = threadIdx.x;
  const int team = blockIdx.x / width, rank = blockIdx.x % width;
  const int Ra = rank == 0 ? 0 : ends[rank - 1], Rb = ends[rank];
  double*   H    = hessians + static_cast<int64_t>(team) * slot;
  double*   v    = vecs + static_cast<int64_t>(blockIdx.x) * (6 + 1 + NW) * n;
  double *  xi = v, *hdg = v + n, *uu = v + 2 * n, *g = v + 3 * n, *diag = v + 4 * n, *part = v + 6 * n;
  for (int i = tid; i < n; i += NT) {
    xi[i]   = 1.0e-3 * ((i * 37 + team) % 101 - 50);
    hdg[i]  = 1.0e-3 * ((i * 53 + 7) % 89 - 44);
    uu[i]   = 1.0e-3 * ((i * 11 + 3) % 97 - 48);
    g[i]    = 1.0e-2 * ((i * 29 + 5) % 83 - 41);
    diag[i] = 1.0;
  }
  for (int64_t i = hess_row_offset(Ra) + tid; i < hess_row_offset(Rb); i += NT) H[i] = 0.0;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    double* lds = reinterpret_cast<double*>(smem);
    hess_pass_rows<true>(diag, H, Ra, Rb, n, true, 1.0e-6, 2.0e-6, 0.5e-3, xi, hdg, uu, g, lds, lds + n, rowsCap, part);
    __syncthreads();
    for (int i = Ra + tid; i < Rb; i += NT) g[i] = 0.5 * g[i] + 1.0e-3 * team_pass_contribution(i, Ra, Rb, rowsCap, lds, lds + n, part);  // the next pass depends on this one
    __syncthreads();
  }
  if (tid == 0) checksum[blockIdx.x] = Rb > Ra ? g[Ra] : 0.0;
}

int main(int argc, char** argv) {
  const int n       = argc > 1 ? std::atoi(argv[1]) : 4252;
  const int width   = argc > 2 ? std::atoi(argv[2]) : 32;
  const int wgs     = argc > 3 ? std::atoi(argv[3]) : 256;
  const int iters   = argc > 4 ? std::atoi(argv[4]) : 20;
  const int rowsCap = argc > 5 ? std::atoi(argv[5]) : 2048;
  const int teams   = wgs / width;
  const int64_t slot = ((hess_row_offset(n) + kHessTailPadDoubles) + 1) & ~int64_t{1};
  std::vector<int> ends(static_cast<size_t>(width));
  for (int q = 0; q < width; ++q) ends[static_cast<size_t>(q)] = row_end(n, width, q);
  double *dH = nullptr, *dV = nullptr, *dSum = nullptr;
  int*    dEnds = nullptr;
  CHECK(hipMalloc(&dH, static_cast<size_t>(slot) * teams * 8));
  CHECK(hipMalloc(&dV, static_cast<size_t>(teams) * width * (7 + NW) * n * 8));
  CHECK(hipMalloc(&dSum, static_cast<size_t>(teams) * width * 8));
  CHECK(hipMalloc(&dEnds, ends.size() * sizeof(int)));
  CHECK(hipMemcpy(dEnds, ends.data(), ends.size() * sizeof(int), hipMemcpyHostToDevice));
  const size_t shmem = static_cast<size_t>(n + team_pass_stage_doubles(std::min(rowsCap, (n + 3) & ~3))) * 8;
  if (shmem > 64 * 1024) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pass_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shmem)));
  auto launch = [&](int its) -> hipError_t {
    hipLaunchKernelGGL(pass_kernel, dim3(teams * width), dim3(NT), shmem, nullptr, dH, slot, dV, n, width, dEnds, its, std::min(rowsCap, (n + 3) & ~3), dSum);
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
  std::vector<double> sums(static_cast<size_t>(teams) * width);
  CHECK(hipMemcpy(sums.data(), dSum, sums.size() * 8, hipMemcpyDeviceToHost));
  const double bytes = static_cast<double>(hess_row_offset(n)) * 16.0 * teams * iters;
  std::printf("{\"threads\": %d, \"n\": %d, \"width\": %d, \"teams\": %d, \"iters\": %d, \"us_per_pass\": %.1f, \"hessian_GBps\": %.0f, \"per_workgroup_GBps\": %.1f, \"checksum\": %.12g}\n",
              NT, n, width, teams, iters, ms * 1e3 / iters, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e9 / (teams * width), sums[0]);
  return 0;
}
