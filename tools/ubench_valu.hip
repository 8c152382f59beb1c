// Microbenchmark: per-instruction VALU issue rates that bound the popcount kernels on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu ; run on the GPU box.
// Each kernel runs N_ITER iterations of a 64-instruction unrolled body with 8 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int N_ITER = 4096;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE> __global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
  uint32_t a[8], b[8];
  int      c[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + i + 1); b[i] = seed ^ (i * 0x9e3779b9u); c[i] = i; }
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if constexpr (MODE == 0) {  // v_and_b32 only (dependent per chain, 8 chains)
#define X(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if constexpr (MODE == 1) {  // v_bcnt_u32_b32 with accumulate
#define X(i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c[i]) : "v"(a[i]));
        REP8(X)
#undef X
      } else if constexpr (MODE == 2) {  // and + bcnt pairs (the kernel's mix): 4 pairs
#define X(i) asm volatile("v_and_b32 %0, %2, %3\n v_bcnt_u32_b32 %1, %0, %1" : "=&v"(b[i]), "+v"(c[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
        X(0) X(1) X(2) X(3)
#undef X
      } else if constexpr (MODE == 3) {  // v_add_u32 (VOP2 reference)
#define X(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if constexpr (MODE == 4) {  // v_fma_f64
        double* d = reinterpret_cast<double*>(a);  // 4 chains of f64
#define X(i) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[i]));
        X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#undef X
      } else if constexpr (MODE == 6) {  // v_cvt_f64_f32 (the dense epilogue converts c, u and the f32 reciprocal seed)
        double* d = reinterpret_cast<double*>(a);
        float*  f = reinterpret_cast<float*>(b);
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
        X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#undef X
      } else if constexpr (MODE == 7) {  // v_rcp_f32
        float* f = reinterpret_cast<float*>(a);
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
        REP8(X)
#undef X
      } else if constexpr (MODE == 8) {  // v_rcp_f64
        double* d = reinterpret_cast<double*>(a);
#define X(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
        X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#undef X
      } else if constexpr (MODE == 9) {  // v_add_f64
        double* d = reinterpret_cast<double*>(a);
#define X(i) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[i]));
        X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#undef X
      } else if constexpr (MODE == 10) {  // v_add_f32
        float* f = reinterpret_cast<float*>(a);
#define X(i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(f[i]));
        REP8(X)
#undef X
      } else if constexpr (MODE == 11) {  // v_cvt_f64_i32
        double* d = reinterpret_cast<double*>(a);
#define X(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(c[i]));
        X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#undef X
      } else if constexpr (MODE == 12) {  // v_rsq_f64
        double* d = reinterpret_cast<double*>(a);
#define X(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));
        X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#undef X
      } else if constexpr (MODE == 13) {  // v_mul_f64
        double* d = reinterpret_cast<double*>(a);
#define X(i) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[i]));
        X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#undef X
      } else if constexpr (MODE == 5) {  // v_and_or_b32 (VOP3 3-src) as a proxy for VOP3 rate
#define X(i) asm volatile("v_and_or_b32 %0, %1, %0, %0" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      }
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < 8; ++i) r += a[i] + b[i] + c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> int run(const char* name, int instrPerBody, int blocksPerCU) {
  int dev = 0; hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, dev));
  const int blocks = p.multiProcessorCount * blocksPerCU;
  uint32_t* d; CHECK(hipMalloc(&d, size_t(blocks) * 256 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double waveInstr = double(blocks) * 4 * N_ITER * 8.0 * instrPerBody;  // per-wave instructions
  const double perCUperClk = waveInstr * 64 / (ms * 1e-3) / p.multiProcessorCount / (p.clockRate * 1e3);
  printf("%-28s blocks/CU=%d  %8.3f ms  %7.2f T lane-ops/s  %6.1f lanes/clk/CU (at %d MHz nominal)\n", name, blocksPerCU, ms,
         waveInstr * 64 / (ms * 1e-3) / 1e12, perCUperClk, p.clockRate / 1000);
  CHECK(hipFree(d));
  return 0;
}

int main() {
  for (int bpc : {4}) {  // the f64 / conversion instructions of the dense-similarity epilogue and the BFGS pair terms
    run<4>("v_fma_f64", 8, bpc);
    run<9>("v_add_f64", 8, bpc);
    run<13>("v_mul_f64", 8, bpc);
    run<6>("v_cvt_f64_f32", 8, bpc);
    run<11>("v_cvt_f64_i32", 8, bpc);
    run<7>("v_rcp_f32", 8, bpc);
    run<8>("v_rcp_f64", 8, bpc);
    run<12>("v_rsq_f64", 8, bpc);
    run<10>("v_add_f32", 8, bpc);
  }
  for (int bpc : {1, 2, 4, 8}) {
    run<0>("v_and_b32", 8, bpc);
    run<1>("v_bcnt_u32_b32 acc", 8, bpc);
    run<2>("and+bcnt pair x4", 8, bpc);
    run<3>("v_add_u32", 8, bpc);
    run<5>("v_and_or_b32 (VOP3)", 8, bpc);
    run<4>("v_fma_f64", 8, bpc);
  }
  return 0;
}
