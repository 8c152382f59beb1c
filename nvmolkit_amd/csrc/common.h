// Shared host-side helpers for libnvmolkit_amd.so: thread-local error slot, HIP error
// checks that return ABI error codes instead of throwing, small RAII wrappers.
// Counterpart of the reference's src/utils/cuda_error_check.h and src/utils/device.h,
// redesigned around a C ABI (no exceptions cross the boundary).
#pragma once

#include <algorithm>

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/nvmolkit_amd.h"

namespace nvmk {

void set_last_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void clear_last_error();

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

template <typename T> inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

// Device scratch that is allocated and released in stream order.
struct StreamScratch {
  void*       ptr    = nullptr;
  hipStream_t stream = nullptr;
  StreamScratch()    = default;
  StreamScratch(const StreamScratch&)            = delete;
  StreamScratch& operator=(const StreamScratch&) = delete;
  hipError_t     alloc(size_t bytes, hipStream_t s) {
    stream = s;
    keep_pool_warm();
    return hipMallocAsync(&ptr, bytes > 0 ? bytes : 1, s);
  }
  // By default the stream-ordered pool hands freed memory back to the driver at every synchronisation, so each call of a
  // blocking entry point re-acquired its scratch from the OS (measured: 40 ms of a 70 ms fused-Butina call at N = 100k).
  // Keep up to 8 GiB cached per device — an eighth of the device's memory where that is more (MI355X: 36 of 288 GB; the team
  // classes of the reference's benchmark file hold several GB of pair records per launch, and a pool that lets go of them at
  // every synchronisation maps them again for every batch: ETKDG 17.8 s in the first whole-file run against 16.0 s in the
  // next); anything beyond is still released.
  static void keep_pool_warm() {
    static bool done[64] = {};
    int         dev     = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || done[dev]) return;
    done[dev] = true;
    hipMemPool_t pool = nullptr;
    if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool != nullptr) {
      uint64_t threshold = 8ull << 30;
      size_t   freeB = 0, totalB = 0;
      if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) threshold = std::max<uint64_t>(threshold, static_cast<uint64_t>(totalB) / 8);
      (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &threshold);
    }
  }
  ~StreamScratch() {
    if (ptr != nullptr) {
      (void)hipFreeAsync(ptr, stream);
    }
  }
  template <typename T> T* as() const { return static_cast<T*>(ptr); }
};

// ---- profiler markers (the reference's ScopedNvtxRange, src/utils/nvtx.h:36-69) ----------------------------------------
// roctx ranges around every C-ABI entry, ETKDG stage and BFGS size class: `rocprofv3 --marker-trace` shows the stage timeline
// of a batch without bespoke scripts.  The roctx library (rocprofiler-sdk's, else roctracer's) is opened lazily with dlopen
// the first time a range is pushed — no link-time dependency, no-ops when it is absent or NVMK_MARKERS=0.
namespace mark {
void push(const char* name);
void pop();
struct Range {
  explicit Range(const char* name) { push(name); }
  ~Range() { pop(); }
  Range(const Range&)            = delete;
  Range& operator=(const Range&) = delete;
};
}  // namespace mark

}  // namespace nvmk

#define NVMK_MARK_CAT2(a, b) a##b
#define NVMK_MARK_CAT(a, b) NVMK_MARK_CAT2(a, b)
#define NVMK_MARK(name) ::nvmk::mark::Range NVMK_MARK_CAT(nvmk_mark_, __LINE__)(name)
#define NVMK_MARK_ENTRY() NVMK_MARK(__func__)

#define NVMK_HIP_CHECK(expr)                                                                              \
  do {                                                                                                    \
    hipError_t nvmk_e_ = (expr);                                                                          \
    if (nvmk_e_ != hipSuccess) {                                                                          \
      ::nvmk::set_last_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(nvmk_e_), __FILE__, __LINE__); \
      return (nvmk_e_ == hipErrorOutOfMemory) ? NVMK_ERR_OUT_OF_MEMORY : NVMK_ERR_HIP;                     \
    }                                                                                                     \
  } while (0)

#define NVMK_REQUIRE(cond, ...)               \
  do {                                        \
    if (!(cond)) {                            \
      ::nvmk::set_last_error(__VA_ARGS__);    \
      return NVMK_ERR_INVALID_ARGUMENT;       \
    }                                         \
  } while (0)

#define NVMK_LAUNCH_CHECK() NVMK_HIP_CHECK(hipGetLastError())
