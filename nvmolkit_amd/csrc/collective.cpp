// Multi-GPU contract of the C ABI (xx_set_devices / xx_allgather_u32; 8(e)): the device set of a process with
// peer access between its members, the asynchronous peer copy that stitches per-GPU results on a target GPU, and the ONE
// collective of the path — the all-gather of the sharded reference fingerprints over RCCL — as thin wrappers a C or C++ host
// (the reference's own Boost.Python modules) can call without torch.
//
// Replaces (reference paths):
//   src/utils/p2p.cpp:30-58        enablePeerAccess (pairwise, idempotent)
//   src/utils/p2p.cpp:60-86        copyDeviceToDeviceAsync (event on the source stream, peer copy on the destination's)
//   src/conformer/device_coord_collector.cpp:86-109   the caller of that copy
// The reference itself has no collective (one process drives all GPUs); configs[4] of BASELINE.json asks for the reference
// block assembled with an RCCL all-gather over xGMI, one process per GPU.
//
// RCCL is reached through dlsym: the library does not link against librccl — a communicator belongs to the RCCL copy that
// made it, and a host process may already carry one (PyTorch ships its own).  The symbols are looked up among the objects
// already loaded first, then in librccl.so.1.
#include <dlfcn.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace {

std::mutex       g_devMutex;
std::vector<int> g_devices;  // empty = every visible device

struct Id128 {  // ncclUniqueId: 128 opaque bytes, passed by value
  char bytes[128];
};
struct Rccl {
  // ncclResult_t is an enum (0 = success), ncclDataType_t ncclUint32 = 3
  int (*getUniqueId)(void*)                                                            = nullptr;
  int (*commInitRank)(void**, int, Id128, int)                                         = nullptr;
  int (*commDestroy)(void*)                                                            = nullptr;
  int (*allGather)(const void*, void*, size_t, int, void*, hipStream_t)                = nullptr;
  const char* (*getErrorString)(int)                                                   = nullptr;
  bool ok                                                                              = false;
};
const Rccl& rccl() {
  static const Rccl r = [] {
    Rccl  x;
    void* handles[4] = {RTLD_DEFAULT, nullptr, nullptr, nullptr};
    int   n          = 1;
    for (const char* lib : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
      if (void* h = dlopen(lib, RTLD_NOW | RTLD_LOCAL)) {
        handles[n++] = h;
        break;
      }
    }
    for (int k = 0; k < n && !x.ok; ++k) {
      void* h = handles[k];
      x.getUniqueId    = reinterpret_cast<decltype(x.getUniqueId)>(dlsym(h, "ncclGetUniqueId"));
      x.commInitRank   = reinterpret_cast<decltype(x.commInitRank)>(dlsym(h, "ncclCommInitRank"));
      x.commDestroy    = reinterpret_cast<decltype(x.commDestroy)>(dlsym(h, "ncclCommDestroy"));
      x.allGather      = reinterpret_cast<decltype(x.allGather)>(dlsym(h, "ncclAllGather"));
      x.getErrorString = reinterpret_cast<decltype(x.getErrorString)>(dlsym(h, "ncclGetErrorString"));
      x.ok             = x.getUniqueId && x.commInitRank && x.commDestroy && x.allGather;
    }
    return x;
  }();
  return r;
}

#define NVMK_RCCL_CHECK(expr)                                                                                     \
  do {                                                                                                            \
    const int rc_ = (expr);                                                                                       \
    if (rc_ != 0) {                                                                                               \
      ::nvmk::set_last_error("%s failed: %s", #expr, rccl().getErrorString ? rccl().getErrorString(rc_) : "RCCL error"); \
      return NVMK_ERR_HIP;                                                                                        \
    }                                                                                                             \
  } while (0)

}  // namespace

extern "C" {

int nvmk_set_devices(const int32_t* device_ids, int n) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(n >= 0 && (n == 0 || device_ids != nullptr), "set_devices: NULL device list");
  int count = 0;
  NVMK_HIP_CHECK(hipGetDeviceCount(&count));
  std::vector<int> ids;
  for (int i = 0; i < n; ++i) {
    NVMK_REQUIRE(device_ids[i] >= 0 && device_ids[i] < count, "set_devices: device %d does not exist (%d visible)", device_ids[i], count);
    for (const int seen : ids) NVMK_REQUIRE(seen != device_ids[i], "set_devices: device %d listed twice", device_ids[i]);
    ids.push_back(device_ids[i]);
  }
  if (ids.empty()) {
    for (int d = 0; d < count; ++d) ids.push_back(d);
  }
  // peer access between every pair, both ways; "already enabled" is the idempotent case (src/utils/p2p.cpp:30-58)
  int current = 0;
  NVMK_HIP_CHECK(hipGetDevice(&current));
  for (const int a : ids) {
    for (const int b : ids) {
      if (a == b) continue;
      int can = 0;
      NVMK_HIP_CHECK(hipDeviceCanAccessPeer(&can, a, b));
      if (!can) continue;  // copies between the two then go through the host, transparently (hipMemcpyPeerAsync)
      NVMK_HIP_CHECK(hipSetDevice(a));
      const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
      if (e == hipErrorPeerAccessAlreadyEnabled) {
        (void)hipGetLastError();
      } else if (e != hipSuccess) {
        (void)hipSetDevice(current);
        ::nvmk::set_last_error("Failed to enable P2P access from GPU %d to GPU %d: %s", a, b, hipGetErrorString(e));
        return NVMK_ERR_HIP;
      }
    }
  }
  NVMK_HIP_CHECK(hipSetDevice(current));
  const std::lock_guard<std::mutex> lock(g_devMutex);
  g_devices = n == 0 ? std::vector<int>() : ids;
  return NVMK_OK;
}

int nvmk_get_devices(int32_t* device_ids, int capacity, int* n) {
  NVMK_REQUIRE(n != nullptr && capacity >= 0 && (capacity == 0 || device_ids != nullptr), "get_devices: NULL argument");
  std::vector<int> ids;
  {
    const std::lock_guard<std::mutex> lock(g_devMutex);
    ids = g_devices;
  }
  if (ids.empty()) {
    int count = 0;
    NVMK_HIP_CHECK(hipGetDeviceCount(&count));
    for (int d = 0; d < count; ++d) ids.push_back(d);
  }
  *n = static_cast<int>(ids.size());
  for (int i = 0; i < *n && i < capacity; ++i) device_ids[i] = ids[static_cast<size_t>(i)];
  return NVMK_OK;
}

int nvmk_copy_peer_async(void* d_dst, int dst_device, void* dst_stream, const void* d_src, int src_device, void* src_stream,
                         size_t bytes) {
  NVMK_MARK_ENTRY();
  if (bytes == 0) return NVMK_OK;
  NVMK_REQUIRE(d_dst != nullptr && d_src != nullptr, "copy_peer: NULL buffer");
  int current = 0;
  NVMK_HIP_CHECK(hipGetDevice(&current));
  // the source stream's work so far must precede the copy: an event recorded there, waited for on the destination stream —
  // also when both buffers live on ONE device (two streams of a device are not ordered either; the reference's same-device branch
  // copies on the destination stream without waiting, src/utils/p2p.cpp:66-70, and leaves the ordering to its caller)
  const bool sameStream = src_device == dst_device && src_stream == dst_stream;
  hipEvent_t ready      = nullptr;
  NVMK_HIP_CHECK(hipSetDevice(src_device));
  hipError_t e = hipSuccess;
  if (!sameStream) {
    e = hipEventCreateWithFlags(&ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ready, nvmk::as_stream(src_stream));
  }
  if (e == hipSuccess) e = hipSetDevice(dst_device);
  if (e == hipSuccess && ready != nullptr) e = hipStreamWaitEvent(nvmk::as_stream(dst_stream), ready, 0);
  if (e == hipSuccess) {
    e = src_device == dst_device ? hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, nvmk::as_stream(dst_stream))
                                 : hipMemcpyPeerAsync(d_dst, dst_device, d_src, src_device, bytes, nvmk::as_stream(dst_stream));
  }
  if (ready != nullptr) (void)hipEventDestroy(ready);  // released once the recorded work has completed
  (void)hipSetDevice(current);
  NVMK_HIP_CHECK(e);
  return NVMK_OK;
}

int nvmk_comm_unique_id(char id[128]) {
  NVMK_REQUIRE(id != nullptr, "comm_unique_id: NULL buffer");
  NVMK_REQUIRE(rccl().ok, "RCCL is not available in this process (librccl.so.1 could not be opened)");
  NVMK_RCCL_CHECK(rccl().getUniqueId(id));
  return NVMK_OK;
}

int nvmk_comm_init_rank(void** comm, int n_ranks, const char id[128], int rank) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(comm != nullptr && id != nullptr, "comm_init_rank: NULL argument");
  NVMK_REQUIRE(n_ranks > 0 && rank >= 0 && rank < n_ranks, "comm_init_rank: rank %d of %d", rank, n_ranks);
  NVMK_REQUIRE(rccl().ok, "RCCL is not available in this process (librccl.so.1 could not be opened)");
  Id128 u;
  for (int i = 0; i < 128; ++i) u.bytes[i] = id[i];
  NVMK_RCCL_CHECK(rccl().commInitRank(comm, n_ranks, u, rank));
  return NVMK_OK;
}

int nvmk_comm_destroy(void* comm) {
  if (comm == nullptr) return NVMK_OK;
  NVMK_REQUIRE(rccl().ok, "RCCL is not available in this process");
  NVMK_RCCL_CHECK(rccl().commDestroy(comm));
  return NVMK_OK;
}

int nvmk_allgather_rows(void* comm, const uint32_t* d_send, int64_t rows_per_rank, int words_per_row, uint32_t* d_recv, void* stream) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(comm != nullptr, "allgather_rows: NULL communicator");
  NVMK_REQUIRE(rows_per_rank >= 0 && words_per_row > 0, "allgather_rows: bad shape (%lld rows of %d words)", (long long)rows_per_rank, words_per_row);
  if (rows_per_rank == 0) return NVMK_OK;
  NVMK_REQUIRE(d_send != nullptr && d_recv != nullptr, "allgather_rows: NULL buffer");
  NVMK_REQUIRE(rccl().ok, "RCCL is not available in this process");
  constexpr int kUint32 = 3;  // ncclUint32
  NVMK_RCCL_CHECK(rccl().allGather(d_send, d_recv, static_cast<size_t>(rows_per_rank) * static_cast<size_t>(words_per_row), kUint32, comm,
                                   nvmk::as_stream(stream)));
  return NVMK_OK;
}

}  // extern "C"
