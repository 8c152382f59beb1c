// Test harness: ONE inverse-Hessian update + next direction through the product's fused pass (nvmolkit_amd/csrc/hess_pass.h, the
// code bfgs_kernel runs), on systems read from a file — the standalone operation the reference tests in
// tests/test_bfgs_hessian.cpp:130-596 (updateInverseHessianBFGSBatch, src/minimizer/bfgs_hessian.cu).  The library has no such
// entry point (its update is fused into the minimisation kernel), so tests/test_hessian_update_gpu.py builds this file once per
// workgroup size and compares its output with the textbook update of oracle/bfgs_update.py.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCHECK_THREADS=64|128|256 tests/native/hess_update_check.hip -o <exe>
//   <exe> in.bin out.bin ldsKB
// in.bin : int32 nSystems, then per system int32 n, int32 active, doubles H[n*n] (symmetric, row-major), dGrad[n], xi[n], grad[n]
// out.bin: per system doubles H[n*n], hessDGrad[n], dGrad[n], xi[n]   (an inactive system comes back as it went in, hessDGrad 0)
// One workgroup per system; rows of the packed triangle live in LDS as far as ldsKB goes and in HBM beyond, as in bfgs_kernel.
// How the two products of the textbook step come out of the pass: pass 1 (no pending update, g = dGrad) gives H dGrad; the
// scalars and u follow; pass 2 applies the update while it forms H_new grad, whose negative is the next direction.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef CHECK_THREADS
#define CHECK_THREADS 256
#endif
#define NVMK_BFGS_NS chk
#define NVMK_BFGS_THREADS CHECK_THREADS
#include "../../nvmolkit_amd/csrc/hess_pass.h"

using namespace nvmk::minim::chk;

#define CHECK(x)                                                          \
  do {                                                                    \
    hipError_t e_ = (x);                                                  \
    if (e_ != hipSuccess) {                                               \
      std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); \
      return 1;                                                           \
    }                                                                     \
  } while (0)

struct Sys {
  int     n, active;
  int64_t full;    // offset of H[n*n] in the dense buffer (input and output)
  int64_t vec;     // offset of this system's vectors (4 n doubles in: dGrad, xi, grad, -; out: hessDGrad, dGrad, xi, -)
  int64_t packed;  // offset of the HBM rows of the packed triangle
};

__global__ __launch_bounds__(NT) void update_kernel(const Sys* __restrict__ systems, double* __restrict__ dense, double* __restrict__ vecs,
                                                    double* __restrict__ packed, const int ldsDoublesMax) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Sys s   = systems[blockIdx.x];
  const int n   = s.n, tid = threadIdx.x;
  if (!s.active) return;
  // the launch's LDS is sized for the largest system; every system uses what bfgs_kernel would give it out of that budget
  const int     ldsDoubles = static_cast<int>(min(static_cast<int64_t>(ldsDoublesMax), lds_vector_doubles(n) + hess_row_offset(n)));
  double*       xi    = reinterpret_cast<double*>(smem);
  double*       hdg   = xi + n;
  double*       uu    = hdg + n;
  double*       g     = uu + n;
  double*       tvec  = g + n;
  double*       dGrad = tvec + n;
  double*       grad  = dGrad + n;
  double*       diag  = xi + 9 * n;  // the layout of bfgs_kernel: 9 vectors, the diagonal, the partial sums, reduction scratch, rows
  double*       part  = diag + n;
  double*       red   = part + (1 + NW) * n;
  double*       Hl    = red + kRedDoubles;
  const int     Rl    = resident_rows(n, lds_hessian_doubles(ldsDoubles, n));
  double*       Hg    = packed + s.packed;
  double*       H     = dense + s.full;
  double*       v     = vecs + s.vec;
  for (int i = tid; i < n; i += NT) {
    dGrad[i] = v[i];
    xi[i]    = v[n + i];
    grad[i]  = v[2 * n + i];
    diag[i]  = H[static_cast<int64_t>(i) * n + i];
    hdg[i] = uu[i] = 0.0;
  }
  for (int r = 1; r < n; ++r) {  // row r of the strict lower triangle: r entries, padded with a zero to an even length
    double* row = r < Rl ? Hl + hess_row_offset(r) : Hg + (hess_row_offset(r) - hess_row_offset(Rl));
    for (int c = tid; c < r + (r & 1); c += NT) row[c] = c < r ? H[static_cast<int64_t>(r) * n + c] : 0.0;
  }
  __syncthreads();
  hess_pass<true>(diag, Hl, Hg, Rl, n, false, 0.0, 0.0, 0.0, xi, hdg, uu, dGrad, part);  // t = H dGrad
  hess_finish(n, part, hdg);
  __syncthreads();
  __shared__ double sc[4];
  if (tid == 0) {  // the scalars in the textbook's own order
    double fac = 0.0, fae = 0.0, sumDGrad = 0.0, sumXi = 0.0;
    for (int i = 0; i < n; ++i) {
      fac += dGrad[i] * xi[i];
      fae += dGrad[i] * hdg[i];
      sumDGrad += dGrad[i] * dGrad[i];
      sumXi += xi[i] * xi[i];
    }
    const bool update = fac > sqrt(3.0e-8 * sumDGrad * sumXi);
    sc[0] = update ? 1.0 : 0.0;
    sc[1] = update ? 1.0 / fac : 0.0;
    sc[2] = update ? 1.0 / fae : 0.0;
    sc[3] = fae;
  }
  __syncthreads();
  const bool   update = sc[0] != 0.0;
  const double rfac = sc[1], fad = sc[2], fae = sc[3];
  for (int i = tid; i < n; i += NT) {
    v[i] = hdg[i];  // hessDGrad out
    if (update) {
      uu[i]    = rfac * xi[i] - fad * hdg[i];
      dGrad[i] = uu[i];
    }
    v[n + i] = dGrad[i];
  }
  __syncthreads();
  hess_pass<true>(diag, Hl, Hg, Rl, n, update, rfac, fad, fae, xi, hdg, uu, grad, part);  // applies the update, t = H_new grad
  hess_finish(n, part, tvec);
  __syncthreads();
  for (int i = tid; i < n; i += NT) {
    v[2 * n + i]                      = -tvec[i];
    H[static_cast<int64_t>(i) * n + i] = diag[i];
  }
  for (int r = 1; r < n; ++r) {
    const double* row = r < Rl ? Hl + hess_row_offset(r) : Hg + (hess_row_offset(r) - hess_row_offset(Rl));
    for (int c = tid; c < r; c += NT) H[static_cast<int64_t>(r) * n + c] = H[static_cast<int64_t>(c) * n + r] = row[c];
  }
}

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s in.bin out.bin ldsKB\n", argv[0]);
    return 2;
  }
  std::FILE* in = std::fopen(argv[1], "rb");
  if (!in) return 2;
  const double ldsKB = std::atof(argv[3]);
  int32_t      nSys  = 0;
  if (std::fread(&nSys, 4, 1, in) != 1) return 2;
  std::vector<Sys>    sys(static_cast<size_t>(nSys));
  std::vector<double> dense, vecs;
  int64_t             packedTotal = 0;
  int                 maxN        = 1;
  for (Sys& s : sys) {
    int32_t hdr[2];
    if (std::fread(hdr, 4, 2, in) != 2) return 2;
    s.n = hdr[0], s.active = hdr[1];
    s.full   = static_cast<int64_t>(dense.size());
    s.vec    = static_cast<int64_t>(vecs.size());
    s.packed = packedTotal;
    const size_t nn = static_cast<size_t>(s.n) * s.n;
    dense.resize(dense.size() + nn);
    vecs.resize(vecs.size() + 4 * static_cast<size_t>(s.n), 0.0);
    if (std::fread(dense.data() + s.full, 8, nn, in) != nn) return 2;
    if (std::fread(vecs.data() + s.vec, 8, 3 * static_cast<size_t>(s.n), in) != 3 * static_cast<size_t>(s.n)) return 2;
    packedTotal += hess_row_offset(s.n) + 2;
    maxN = std::max(maxN, s.n);
  }
  std::fclose(in);
  const size_t vecBytes = static_cast<size_t>(lds_vector_doubles(maxN)) * 8;
  size_t       shmem    = std::max<size_t>(vecBytes, static_cast<size_t>(ldsKB * 1024) & ~size_t{15});
  shmem                 = std::min(shmem, vecBytes + static_cast<size_t>(hess_row_offset(maxN)) * 8);
  if (shmem > 160 * 1024 - 64) {
    std::fprintf(stderr, "system of %d coordinates needs %zu bytes of LDS\n", maxN, shmem);
    return 3;
  }
  Sys*    dSys = nullptr;
  double *dDense = nullptr, *dVecs = nullptr, *dPacked = nullptr;
  CHECK(hipMalloc(&dSys, sys.size() * sizeof(Sys)));
  CHECK(hipMalloc(&dDense, std::max<size_t>(dense.size(), 1) * 8));
  CHECK(hipMalloc(&dVecs, std::max<size_t>(vecs.size(), 1) * 8));
  CHECK(hipMalloc(&dPacked, static_cast<size_t>(packedTotal + kHessTailPadDoubles) * 8));
  CHECK(hipMemset(dPacked, 0, static_cast<size_t>(packedTotal + kHessTailPadDoubles) * 8));
  CHECK(hipMemcpy(dSys, sys.data(), sys.size() * sizeof(Sys), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dDense, dense.data(), dense.size() * 8, hipMemcpyHostToDevice));
  // the output layout of a system's vectors is (hessDGrad, dGrad, xi): the kernel reads (dGrad, xi, grad) first
  CHECK(hipMemcpy(dVecs, vecs.data(), vecs.size() * 8, hipMemcpyHostToDevice));
  if (shmem > 64 * 1024)
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(update_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shmem)));
  hipLaunchKernelGGL(update_kernel, dim3(nSys), dim3(NT), shmem, nullptr, dSys, dDense, dVecs, dPacked, static_cast<int>(shmem / 8));
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  std::vector<double> outVecs(vecs.size());
  CHECK(hipMemcpy(dense.data(), dDense, dense.size() * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(outVecs.data(), dVecs, outVecs.size() * 8, hipMemcpyDeviceToHost));
  std::FILE* out = std::fopen(argv[2], "wb");
  if (!out) return 2;
  for (const Sys& s : sys) {
    const size_t n = static_cast<size_t>(s.n);
    std::fwrite(dense.data() + s.full, 8, n * n, out);
    if (s.active) {
      std::fwrite(outVecs.data() + s.vec, 8, 3 * n, out);
    } else {  // untouched: hessDGrad 0, dGrad and xi as they came
      std::vector<double> zero(n, 0.0);
      std::fwrite(zero.data(), 8, n, out);
      std::fwrite(vecs.data() + s.vec, 8, 2 * n, out);
    }
  }
  std::fclose(out);
  std::printf("threads %d systems %d largest %d lds %zu\n", NT, nSys, maxN, shmem);
  return 0;
}
