// Library-level entry points: error slot, ABI version, device queries.
#include <cstring>
#include <dlfcn.h>

#include <initializer_list>
#include <mutex>

#include "common.h"
#include "options.h"

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

namespace opt {
namespace {
const char* const kNames[kNumOptions] = {"NVMK_SIM_PATH",       "NVMK_COUNT_THRESHOLD", "NVMK_COUNT_SUPER",  "NVMK_COUNT_KERNEL", "NVMK_BUTINA_ROUNDS",
                                         "NVMK_BUTINA_SORT",    "NVMK_BFGS_LDS",        "NVMK_BFGS_XCD_GROUP", "NVMK_BFGS_PROFILE",
                                         "NVMK_BFGS_VECTORS",   "NVMK_BFGS_OVERLAP",    "NVMK_BFGS_WAVE",   "NVMK_BFGS_WAVE2",
                                         "NVMK_BFGS_WAVE8",     "NVMK_BFGS_HESS_CAP_MB", "NVMK_BFGS_TIMELINE", "NVMK_BFGS_SCHED", "NVMK_MARKERS", "NVMK_ETKDG_TIMING", "NVMK_ETKDG_PRUNE",
                                         "NVMK_BUILD_SLOT_KB",  "NVMK_BFGS_TEAM",       "NVMK_BFGS_TEAM_WIDTH", "NVMK_BFGS_TEAM_SHARE_KB",
                                         "NVMK_BFGS_TEAM_THREADS", "NVMK_BFGS_TEAM_TIMEOUT_MS", "NVMK_BFGS_HISTORY"};
std::mutex g_mutex;
Text       g_values[kNumOptions];
bool       g_loaded = false;
void store(Text& t, const char* v) {
  std::memset(t.s, 0, sizeof(t.s));
  if (v != nullptr) std::strncpy(t.s, v, sizeof(t.s) - 1);
}
void load_locked() {  // the one place the environment is read
  if (g_loaded) return;
  for (int i = 0; i < kNumOptions; ++i) store(g_values[i], std::getenv(kNames[i]));
  g_loaded = true;
}
}  // namespace
const char* name(const Id id) { return kNames[id]; }
int find(const char* n) {
  if (n == nullptr) return -1;
  for (int i = 0; i < kNumOptions; ++i)
    if (std::strcmp(kNames[i], n) == 0) return i;
  return -1;
}
Text get(const Id id) {
  std::lock_guard<std::mutex> lock(g_mutex);
  load_locked();
  return g_values[id];
}
int set(const char* n, const char* value) {
  const int id = find(n);
  if (id < 0) return -1;
  std::lock_guard<std::mutex> lock(g_mutex);
  load_locked();
  store(g_values[id], value);
  return 0;
}
}  // namespace opt

namespace mark {
namespace {
using PushFn = int (*)(const char*);
using PopFn  = int (*)();
struct Api {
  PushFn push = nullptr;
  PopFn  pop  = nullptr;
};
const Api& api() {
  static const Api a = [] {
    Api x;
    if (opt::get(opt::kMarkers).is("0")) return x;
    for (const char* lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      if (void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) {
        x.push = reinterpret_cast<PushFn>(dlsym(h, "roctxRangePushA"));
        x.pop  = reinterpret_cast<PopFn>(dlsym(h, "roctxRangePop"));
        if (x.push != nullptr && x.pop != nullptr) return x;
        x = Api{};
      }
    }
    return x;
  }();
  return a;
}
}  // namespace
void push(const char* name) {
  if (const PushFn f = api().push) (void)f(name);
}
void pop() {
  if (const PopFn f = api().pop) (void)f();
}
}  // namespace mark

}  // namespace nvmk

extern "C" {

const char* nvmk_last_error(void) { return nvmk::g_last_error; }

int nvmk_abi_version(void) { return (0 << 16) | 4; }

int nvmk_set_option(const char* name, const char* value) {
  NVMK_REQUIRE(name != nullptr, "nvmk_set_option: name is NULL");
  NVMK_REQUIRE(value == nullptr || std::strlen(value) < sizeof(nvmk::opt::Text::s), "nvmk_set_option: value too long");
  NVMK_REQUIRE(nvmk::opt::set(name, value) == 0, "nvmk_set_option: unknown option '%s'", name);
  return NVMK_OK;
}

int nvmk_get_option(const char* name, char* value, size_t capacity) {
  NVMK_REQUIRE(name != nullptr && value != nullptr && capacity > 0, "nvmk_get_option: NULL argument");
  const int id = nvmk::opt::find(name);
  NVMK_REQUIRE(id >= 0, "nvmk_get_option: unknown option '%s'", name);
  const nvmk::opt::Text t = nvmk::opt::get(static_cast<nvmk::opt::Id>(id));
  std::strncpy(value, t.s, capacity - 1);
  value[capacity - 1] = '\0';
  return NVMK_OK;
}

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
