// ThreadSanitizer driver of the option registry (nvmolkit_amd/csrc/runtime.cpp): include/nvmolkit_amd.h promises that
// nvmk_set_option is safe against concurrent callers of the other entry points.  Writers flip two switches while readers take
// snapshots through nvmk_get_option; every value a reader sees must be one a writer wrote, whole.  Built with
// -fsanitize=thread by tests/test_options_threads.py; a data-race report or a torn value fails the test.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "nvmolkit_amd.h"

int main() {
  static const char* kValues[] = {"auto", "mfma", "valu", ""};
  std::atomic<bool> stop{false};
  std::atomic<long> torn{0}, reads{0};
  std::vector<std::thread> pool;
  for (int w = 0; w < 2; ++w)
    pool.emplace_back([&, w] {
      for (long i = 0; !stop.load(); ++i) {
        nvmk_set_option("NVMK_SIM_PATH", kValues[(i + w) % 4]);
        nvmk_set_option("NVMK_BFGS_LDS", (i & 1) ? "full" : "auto");
      }
    });
  for (int r = 0; r < 4; ++r)
    pool.emplace_back([&] {
      char buf[64];
      for (long i = 0; i < 200000; ++i) {
        if (nvmk_get_option("NVMK_SIM_PATH", buf, sizeof buf) != 0) { ++torn; continue; }
        bool ok = false;
        for (const char* v : kValues) ok = ok || std::strcmp(buf, v) == 0;
        if (!ok) ++torn;
        if (nvmk_get_option("NVMK_BFGS_LDS", buf, sizeof buf) != 0 || (std::strcmp(buf, "full") != 0 && std::strcmp(buf, "auto") != 0 && buf[0] != '\0')) ++torn;
        ++reads;
      }
    });
  for (size_t t = 2; t < pool.size(); ++t) pool[t].join();
  stop.store(true);
  pool[0].join();
  pool[1].join();
  if (nvmk_set_option("NVMK_NO_SUCH_SWITCH", "1") == 0) { std::printf("unknown name accepted\n"); return 2; }
  std::printf("reads %ld torn %ld\n", reads.load(), torn.load());
  return torn.load() == 0 ? 0 : 1;
}
