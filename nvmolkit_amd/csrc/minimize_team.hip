// The cooperative BFGS class — several workgroups per system — as a translation unit of its own (the kernels of minimize.hip take
// minutes to compile; these compile beside them).  Device code: bfgs_device.inc (Team, bfgs_system<..., TEAM = true>,
// bfgs_team_kernel) and hess_pass.h (hess_pass_rows); host side (which systems form a team class, memory, launch order):
// minimize.hip, which calls launch_team_kernel below.
//
// Replaces (reference paths): src/minimizer/bfgs_minimize_permol_kernels.cu:796-932 — the reference sends systems that do not
// fit shared memory to global-memory instantiations (MaxAtoms 256 / 2048) of its one-block-per-molecule kernel, and its
// benchmark (benchmarks/etkdg_bench.py:193) feeds the whole chembl_10k.smi, peptides of up to 1063 atoms included.
#include <cstdio>

#include "bfgs_common.h"

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

namespace nvmk {
namespace minim {

int launch_team_kernel(const int threads, const bool profile, const unsigned grid, const size_t shmem, hipStream_t stream,
                       const Batch& b, const BfgsArgs& A) {
  auto go = [&](auto kern) -> int {
    if (shmem > 64 * 1024) {
      NVMK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(shmem)));
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(static_cast<unsigned>(threads)), shmem, stream, b, A);
    NVMK_LAUNCH_CHECK();
    return NVMK_OK;
  };
  int r = NVMK_OK;
  if (profile && (b.kind == NVMK_FF_DG || b.kind == NVMK_FF_ETK || b.kind == NVMK_FF_MMFF)) {
    if (threads == 512) {
      if (b.kind == NVMK_FF_DG) return go(t512::bfgs_team_kernel<NVMK_FF_DG, true>);
      if (b.kind == NVMK_FF_ETK) return go(t512::bfgs_team_kernel<NVMK_FF_ETK, true>);
      return go(t512::bfgs_team_kernel<NVMK_FF_MMFF, true>);
    }
    if (b.kind == NVMK_FF_DG) return go(t256::bfgs_team_kernel<NVMK_FF_DG, true>);
    if (b.kind == NVMK_FF_ETK) return go(t256::bfgs_team_kernel<NVMK_FF_ETK, true>);
    return go(t256::bfgs_team_kernel<NVMK_FF_MMFF, true>);
  }
  if (threads == 512) {
    NVMK_FF_DISPATCH(b.kind, r = go(t512::bfgs_team_kernel<K>));
  } else {
    NVMK_FF_DISPATCH(b.kind, r = go(t256::bfgs_team_kernel<K>));
  }
  return r;
}

}  // namespace minim
}  // namespace nvmk
