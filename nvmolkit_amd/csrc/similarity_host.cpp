// Memory-constrained cross-similarity returned on the host.
//
// Replaces crossSimilarityImpl / cross{Tanimoto,Cosine}SimilarityCPUResult
// (reference: src/similarity.cpp:105-254, :282-297).  Same contract: if the N x M double matrix fits
// in the allowed device memory it is computed in one launch and copied back; otherwise rows of the
// first operand are processed in chunks of max(32, floor(((allowed/2)*0.9/8) / (32*M)) * 32) rows on two
// streams, and NVMK_ERR_OUT_OF_MEMORY is returned when even 32 rows do not fit (src/similarity.cpp:126-139).
//
// Host-side design: two worker threads (std::thread, no OpenMP runtime inside the library), each
// owning a stream, one device chunk buffer and two pinned staging buffers; D2H of piece k+1 overlaps
// the pageable memcpy of piece k.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>

#include "common.h"
#include "options.h"

namespace {

constexpr size_t kPinnedBytes = 64ULL << 20;  // per staging buffer (reference uses 100 MB, src/similarity.cpp:146)

struct Worker {
  hipStream_t stream   = nullptr;
  double*     dBuf     = nullptr;
  double*     pinned[2] = {nullptr, nullptr};
  hipEvent_t  ev[2]    = {nullptr, nullptr};
  int         rc       = NVMK_OK;
  char        err[512] = {0};

  ~Worker() {
    for (int k = 0; k < 2; ++k) {
      if (pinned[k]) (void)hipHostFree(pinned[k]);
      if (ev[k]) (void)hipEventDestroy(ev[k]);
    }
    if (dBuf) (void)hipFree(dBuf);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

int launch_metric(int metric, const uint32_t* a, int64_t nA, const uint32_t* b, int64_t nB, int fpBits, double* out,
                  hipStream_t s) {
  return metric == NVMK_METRIC_TANIMOTO ? nvmk_cross_tanimoto_f64(a, nA, b, nB, fpBits, out, nB, s) :
                                          nvmk_cross_cosine_f64(a, nA, b, nB, fpBits, out, nB, s);
}

// Copy `count` doubles from device `src` to pageable `dst` through the worker's two pinned buffers.
int drain_to_host(Worker& w, const double* src, double* dst, size_t count) {
  const size_t cap     = kPinnedBytes / sizeof(double);
  size_t       prevOff = 0, prevLen = 0;
  int          prevBuf = -1;
  int          buf     = 0;
  for (size_t off = 0; off < count; off += cap, buf ^= 1) {
    const size_t len = std::min(cap, count - off);
    NVMK_HIP_CHECK(hipMemcpyAsync(w.pinned[buf], src + off, len * sizeof(double), hipMemcpyDeviceToHost, w.stream));
    NVMK_HIP_CHECK(hipEventRecord(w.ev[buf], w.stream));
    if (prevBuf >= 0) {
      NVMK_HIP_CHECK(hipEventSynchronize(w.ev[prevBuf]));
      std::memcpy(dst + prevOff, w.pinned[prevBuf], prevLen * sizeof(double));
    }
    prevOff = off;
    prevLen = len;
    prevBuf = buf;
  }
  if (prevBuf >= 0) {
    NVMK_HIP_CHECK(hipEventSynchronize(w.ev[prevBuf]));
    std::memcpy(dst + prevOff, w.pinned[prevBuf], prevLen * sizeof(double));
  }
  return NVMK_OK;
}

int init_worker(Worker& w, size_t chunkDoubles) {
  NVMK_HIP_CHECK(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
  NVMK_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&w.dBuf), chunkDoubles * sizeof(double)));
  const size_t pinBytes = std::min(kPinnedBytes, chunkDoubles * sizeof(double));
  for (int k = 0; k < 2; ++k) {
    NVMK_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&w.pinned[k]), pinBytes, hipHostMallocDefault));
    NVMK_HIP_CHECK(hipEventCreateWithFlags(&w.ev[k], hipEventDisableTiming));
  }
  return NVMK_OK;
}

}  // namespace

extern "C" int nvmk_cross_similarity_host_f64(int metric, const uint32_t* d_a, int64_t nA, const uint32_t* d_b,
                                              int64_t nB, int fp_bits, double* h_out, int64_t max_device_bytes) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(metric == NVMK_METRIC_TANIMOTO || metric == NVMK_METRIC_COSINE, "unknown metric %d", metric);
  NVMK_REQUIRE(fp_bits > 0 && fp_bits % 32 == 0, "fp_bits must be a positive multiple of 32, got %d", fp_bits);
  NVMK_REQUIRE(nA >= 0 && nB >= 0, "negative row count");
  if (nA == 0 || nB == 0) {
    return NVMK_OK;
  }
  NVMK_REQUIRE(d_a && d_b && h_out, "NULL buffer");
  size_t allowed = 0;
  if (max_device_bytes < 0) {
    size_t total = 0;
    NVMK_HIP_CHECK(hipMemGetInfo(&allowed, &total));
  } else {
    allowed = static_cast<size_t>(max_device_bytes);
  }
  int device = 0;
  NVMK_HIP_CHECK(hipGetDevice(&device));

  const size_t total    = static_cast<size_t>(nA) * static_cast<size_t>(nB);
  const int    W        = fp_bits / 32;
  size_t       chunkRows;
  int          nWorkers;
  // The single-chunk path needs the whole result AND the FP4-expanded operand sets the launch allocates (fp4.h: 16 bytes
  // per fingerprint word) on the device, with the same 10 % headroom the chunked path keeps: a matrix that only just fits
  // the free memory goes through the chunked path instead of failing with out-of-memory.
  const size_t operandBytes = nvmk_fp4_workspace_bytes(nA, fp_bits) + nvmk_fp4_workspace_bytes(nB, fp_bits);
  if (allowed / 10 * 9 >= total * sizeof(double) + operandBytes) {
    chunkRows = static_cast<size_t>(nA);
    nWorkers  = 1;
  } else {
    // reference: src/similarity.cpp:130-140
    const size_t perBuffer   = allowed / 2;
    const size_t maxDoubles  = (perBuffer / sizeof(double) * 9) / 10;
    const size_t minRows     = 32;
    const size_t increment   = minRows * static_cast<size_t>(nB);
    if (increment > maxDoubles) {
      nvmk::set_last_error("Not enough memory to compute cross similarity (32 x %lld doubles do not fit in %zu bytes)",
                           (long long)nB, allowed);
      return NVMK_ERR_OUT_OF_MEMORY;
    }
    chunkRows = std::max(minRows, (maxDoubles / increment) * minRows);
    nWorkers  = 2;
  }
  // Several chunks on the matrix-core path: both operands are expanded to FP4 ONCE (nvmk_fp4_prepare) and every chunk is a launch
  // on the prepared sets — nvmk_cross_tanimoto_f64 would expand the whole of B again for every chunk of rows.  The
  // prepared launch wants row offsets that are multiples of 128, so the chunk is rounded down to one (still within the
  // reference's memory bound); chunks too small for that path keep the per-chunk entry point.
  struct Prepared {
    void* a = nullptr;
    void* b = nullptr;
    ~Prepared() {
      if (a) (void)hipFree(a);
      if (b) (void)hipFree(b);
    }
  } prepared;
  bool usePrepared = false;
  if (nWorkers == 2 && chunkRows >= 128 && W >= 4 && nB >= 64 && !nvmk::opt::get(nvmk::opt::kSimPath).is("valu") &&
      static_cast<double>(chunkRows / 128 * 128) * static_cast<double>(nB) >= 4.0e6) {
    chunkRows = chunkRows / 128 * 128;
    if (static_cast<size_t>(nA) > chunkRows) {
      NVMK_HIP_CHECK(hipMalloc(&prepared.a, nvmk_fp4_workspace_bytes(nA, fp_bits)));
      NVMK_HIP_CHECK(hipMalloc(&prepared.b, nvmk_fp4_workspace_bytes(nB, fp_bits)));
      int rc = nvmk_fp4_prepare(d_a, nA, fp_bits, prepared.a, nullptr);
      if (rc == NVMK_OK) rc = nvmk_fp4_prepare(d_b, nB, fp_bits, prepared.b, nullptr);
      if (rc != NVMK_OK) return rc;
      NVMK_HIP_CHECK(hipStreamSynchronize(nullptr));  // the workers' streams do not wait for the null stream
      usePrepared = true;
    }
  }
  const size_t nChunks = (static_cast<size_t>(nA) + chunkRows - 1) / chunkRows;
  nWorkers             = static_cast<int>(std::min<size_t>(nWorkers, nChunks));

  Worker              workers[2];
  std::atomic<size_t> next{0};
  auto                body = [&](int wi) {
    Worker& w = workers[wi];
    if (hipSetDevice(device) != hipSuccess) {
      w.rc = NVMK_ERR_HIP;
      snprintf(w.err, sizeof(w.err), "hipSetDevice(%d) failed in worker", device);
      return;
    }
    w.rc = init_worker(w, std::min(chunkRows, static_cast<size_t>(nA)) * static_cast<size_t>(nB));
    while (w.rc == NVMK_OK) {
      const size_t c = next.fetch_add(1);
      if (c >= nChunks) break;
      const size_t row0 = c * chunkRows;
      const size_t rows = std::min(chunkRows, static_cast<size_t>(nA) - row0);
      w.rc = usePrepared ? nvmk_cross_similarity_prepared_f64(metric, prepared.a, nA, static_cast<int64_t>(row0), static_cast<int64_t>(rows),
                                                              prepared.b, nB, fp_bits, w.dBuf, nB, w.stream)
                         : launch_metric(metric, d_a + row0 * W, static_cast<int64_t>(rows), d_b, nB, fp_bits, w.dBuf, w.stream);
      if (w.rc != NVMK_OK) break;
      w.rc = drain_to_host(w, w.dBuf, h_out + row0 * static_cast<size_t>(nB), rows * static_cast<size_t>(nB));
    }
    if (w.rc != NVMK_OK) {
      snprintf(w.err, sizeof(w.err), "%s", nvmk_last_error());  // error slot is thread-local: carry it out
    } else if (hipStreamSynchronize(w.stream) != hipSuccess) {
      w.rc = NVMK_ERR_HIP;
      snprintf(w.err, sizeof(w.err), "hipStreamSynchronize failed in worker");
    }
  };
  if (nWorkers == 1) {
    body(0);
  } else {
    std::thread t1(body, 1);
    body(0);
    t1.join();
  }
  for (int wi = 0; wi < nWorkers; ++wi) {
    if (workers[wi].rc != NVMK_OK) {
      nvmk::set_last_error("%s", workers[wi].err);
      return workers[wi].rc;
    }
  }
  return NVMK_OK;
}
