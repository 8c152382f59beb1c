// BASELINE.json configs[4] from a C++ host, one process per GPU: the query rows are this rank's own, the reference set is
// sharded over the ranks and assembled with ONE all-gather over RCCL (nvmk_allgather_rows) in front of the similarity launch,
// on the same stream.  No torch, nothing of the library's C++ crosses the boundary.
//   sharded_reference <rank> <n_ranks> <id_file>      (rank 0 writes the 128-byte communicator id to id_file, the others wait
//                                                      for it; with no arguments: one rank, which is what a one-GPU box can run)
// Every rank checks its block of the result against a host popcount loop over the WHOLE reference set (bit for bit).
//   g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/sharded_reference_from_cxx.cpp \
//       -Lnvmolkit_amd/lib -lnvmolkit_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/nvmolkit_amd/lib -Wl,-rpath,/opt/rocm/lib -o sharded_reference
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
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
#define NVMK_DO(call)                                         \
  do {                                                        \
    if ((call) != NVMK_OK) {                                  \
      std::fprintf(stderr, "%s: %s\n", #call, nvmk_last_error()); \
      return 1;                                               \
    }                                                         \
  } while (0)

int main(int argc, char** argv) {
  const int   rank = argc > 1 ? std::atoi(argv[1]) : 0, nRanks = argc > 2 ? std::atoi(argv[2]) : 1;
  const char* idFile = argc > 3 ? argv[3] : nullptr;
  int         nDev = 0;
  NVMK_DO(nvmk_device_count(&nDev));
  HIP_OK(hipSetDevice(rank % (nDev > 0 ? nDev : 1)));

  char id[128] = {0};
  if (rank == 0) {
    NVMK_DO(nvmk_comm_unique_id(id));
    if (idFile) {
      std::FILE* f = std::fopen(idFile, "wb");
      if (!f || std::fwrite(id, 1, 128, f) != 128) return 4;
      std::fclose(f);
    }
  } else {
    for (int tries = 0;; ++tries) {
      std::FILE* f = idFile ? std::fopen(idFile, "rb") : nullptr;
      const size_t got = f ? std::fread(id, 1, 128, f) : 0;
      if (f) std::fclose(f);
      if (got == 128) break;
      if (tries > 3000) return 4;
      std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
  }
  void* comm = nullptr;
  NVMK_DO(nvmk_comm_init_rank(&comm, nRanks, id, rank));

  // the same seeded sets on every rank, so that each can check its block on the host
  const int64_t nQ = 300, rowsPerRank = 256, nRef = rowsPerRank * nRanks;
  const int     bits = 2048, words = bits / 32;
  std::mt19937  rng(11);
  std::vector<uint32_t> ref(nRef * words), q(nQ * nRanks * words);
  for (auto& w : ref) w = rng() & rng() & rng();
  for (auto& w : q) w = rng() & rng() & rng();
  const uint32_t* myQ   = q.data() + static_cast<size_t>(rank) * nQ * words;
  const uint32_t* myRef = ref.data() + static_cast<size_t>(rank) * rowsPerRank * words;

  uint32_t *dQ = nullptr, *dShard = nullptr, *dAll = nullptr;
  double*   dOut = nullptr;
  hipStream_t stream = nullptr;
  HIP_OK(hipStreamCreate(&stream));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&dQ), nQ * words * sizeof(uint32_t)));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&dShard), rowsPerRank * words * sizeof(uint32_t)));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&dAll), nRef * words * sizeof(uint32_t)));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&dOut), nQ * nRef * sizeof(double)));
  HIP_OK(hipMemcpyAsync(dQ, myQ, nQ * words * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(dShard, myRef, rowsPerRank * words * sizeof(uint32_t), hipMemcpyHostToDevice, stream));

  NVMK_DO(nvmk_allgather_rows(comm, dShard, rowsPerRank, words, dAll, stream));                 // the one collective
  NVMK_DO(nvmk_cross_tanimoto_f64(dQ, nQ, dAll, nRef, bits, dOut, /*ld_out=*/nRef, stream));   // this rank's rows of the matrix
  std::vector<double> out(nQ * nRef);
  HIP_OK(hipMemcpyAsync(out.data(), dOut, out.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));

  int64_t wrong = 0;
  for (int64_t i = 0; i < nQ; ++i) {
    for (int64_t j = 0; j < nRef; ++j) {
      int c = 0, pa = 0, pb = 0;
      for (int w = 0; w < words; ++w) {
        c += __builtin_popcount(myQ[i * words + w] & ref[j * words + w]);
        pa += __builtin_popcount(myQ[i * words + w]);
        pb += __builtin_popcount(ref[j * words + w]);
      }
      const int u = pa + pb - c;
      wrong += (out[i * nRef + j] != static_cast<double>(c) / static_cast<double>(u > 1 ? u : 1));
    }
  }
  std::printf("rank %d of %d: %lld x %lld similarities against the gathered reference set, %lld differ from the host loop\n", rank, nRanks,
              (long long)nQ, (long long)nRef, (long long)wrong);
  NVMK_DO(nvmk_comm_destroy(comm));
  HIP_OK(hipFree(dQ));
  HIP_OK(hipFree(dShard));
  HIP_OK(hipFree(dAll));
  HIP_OK(hipFree(dOut));
  HIP_OK(hipStreamDestroy(stream));
  return wrong == 0 ? 0 : 3;
}
