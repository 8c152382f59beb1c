// Library-level entry points: error slot, ABI version, device queries.
#include <cstring>

#include "common.h"

namespace nvmk {

namespace {
thread_local char g_last_error[1024] = {0};
}

void set_last_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

void clear_last_error() { g_last_error[0] = '\0'; }

}  // namespace nvmk

extern "C" {

const char* nvmk_last_error(void) { return nvmk::g_last_error; }

int nvmk_abi_version(void) { return (0 << 16) | 3; }

int nvmk_device_count(int* count) {
  NVMK_REQUIRE(count != nullptr, "nvmk_device_count: count is NULL");
  NVMK_HIP_CHECK(hipGetDeviceCount(count));
  return NVMK_OK;
}

int nvmk_device_memory(size_t* free_bytes, size_t* total_bytes) {
  size_t f = 0, t = 0;
  NVMK_HIP_CHECK(hipMemGetInfo(&f, &t));
  if (free_bytes != nullptr) *free_bytes = f;
  if (total_bytes != nullptr) *total_bytes = t;
  return NVMK_OK;
}

}  // extern "C"
