// A C++ caller of libnvmolkit_amd.so — the call the reference's src/similarity.cpp:38-58 (crossTanimotoSimilarityGpuResult)
// would make in place of launching src/similarity_kernels.cu:505-582: device pointers, sizes and a stream go in, an error code
// comes out; nothing of torch or of this library's C++ crosses the boundary.  The result is checked against a host popcount loop
// (same integers, one IEEE double division: bit for bit).
//   g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/cross_tanimoto_from_cxx.cpp \
//       -Lnvmolkit_amd/lib -lnvmolkit_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/nvmolkit_amd/lib -Wl,-rpath,/opt/rocm/lib -o cross_tanimoto
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

extern "C" {
#include "nvmolkit_amd.h"
}

#define HIP_OK(call)                                                                      \
  do {                                                                                    \
    const hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess) {                                                               \
      std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                     \
      return 2;                                                                           \
    }                                                                                     \
  } while (0)

int main() {
  const int64_t nA = 1000, nB = 777;
  const int     bits = 2048, words = bits / 32;
  std::mt19937  rng(7);
  std::vector<uint32_t> a(nA * words), b(nB * words);
  for (auto& w : a) w = rng() & rng() & rng();  // ~12 % density
  for (auto& w : b) w = rng() & rng() & rng();

  uint32_t *dA = nullptr, *dB = nullptr;
  double*   dOut = nullptr;
  hipStream_t stream = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&dA), a.size() * sizeof(uint32_t)));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&dB), b.size() * sizeof(uint32_t)));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&dOut), nA * nB * sizeof(double)));
  HIP_OK(hipStreamCreate(&stream));
  HIP_OK(hipMemcpyAsync(dA, a.data(), a.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(dB, b.data(), b.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));

  const int rc = nvmk_cross_tanimoto_f64(dA, nA, dB, nB, bits, dOut, /*ld_out=*/nB, stream);
  if (rc != NVMK_OK) {
    std::fprintf(stderr, "nvmk_cross_tanimoto_f64: %s\n", nvmk_last_error());
    return 1;
  }
  std::vector<double> out(nA * nB);
  HIP_OK(hipMemcpyAsync(out.data(), dOut, out.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));

  int64_t wrong = 0;
  for (int64_t i = 0; i < nA; ++i) {
    for (int64_t j = 0; j < nB; ++j) {
      int c = 0, pa = 0, pb = 0;
      for (int w = 0; w < words; ++w) {
        c += __builtin_popcount(a[i * words + w] & b[j * words + w]);
        pa += __builtin_popcount(a[i * words + w]);
        pb += __builtin_popcount(b[j * words + w]);
      }
      const int    u    = pa + pb - c;
      const double want = static_cast<double>(c) / static_cast<double>(u > 1 ? u : 1);
      wrong += (out[i * nB + j] != want);
    }
  }
  std::printf("%lld x %lld Tanimoto similarities, %lld differ from the host loop\n", (long long)nA, (long long)nB, (long long)wrong);
  HIP_OK(hipFree(dA));
  HIP_OK(hipFree(dB));
  HIP_OK(hipFree(dOut));
  HIP_OK(hipStreamDestroy(stream));
  return wrong == 0 ? 0 : 3;
}
