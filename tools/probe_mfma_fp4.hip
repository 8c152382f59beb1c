// Probe: does v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (e2m1) operands compute exact popcount(a&b)?
// One wave: 32 "A" fingerprints x 32 "B" fingerprints of 64 bits each, expanded to fp4 nibbles
// (bit -> 0x2 = 1.0 in e2m1, else 0x0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <chrono>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// expand 64 bits -> 32 bytes (64 nibbles); nibble k = bit k
__host__ __device__ inline void expand64(uint64_t bits, uint8_t* out32) {
  for (int k = 0; k < 32; ++k) {
    uint8_t lo = ((bits >> (2 * k)) & 1) ? 0x2 : 0x0;
    uint8_t hi = ((bits >> (2 * k + 1)) & 1) ? 0x2 : 0x0;
    out32[k]   = lo | (hi << 4);
  }
}

template <int SCALE>
__global__ void probe(const uint8_t* A, const uint8_t* B, float* D) {
  const int lane = threadIdx.x;
  // lane l: row l&31, k half l>>5: 16 bytes = 32 fp4
  v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  const int4 av = *reinterpret_cast<const int4*>(A + (lane & 31) * 32 + (lane >> 5) * 16);
  const int4 bv = *reinterpret_cast<const int4*>(B + (lane & 31) * 32 + (lane >> 5) * 16);
  a[0] = av.x; a[1] = av.y; a[2] = av.z; a[3] = av.w;
  b[0] = bv.x; b[1] = bv.y; b[2] = bv.z; b[3] = bv.w;
  v16f c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, SCALE, 0, SCALE);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int col = lane & 31;
    D[row * 32 + col] = c[r];
  }
}

// throughput: 64 MFMAs per iteration, 4 independent accumulators
template <int SCALE>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
  v8i a = {0x22222222, 0x22222222, 0x22222222, 0x22222222, 0, 0, 0, 0};
  v8i b = a;
  a[0] ^= threadIdx.x;
  v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, SCALE, 0, SCALE);
      c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, SCALE, 0, SCALE);
      c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 4, 4, 0, SCALE, 0, SCALE);
      c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 4, 4, 0, SCALE, 0, SCALE);
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SCALE> int run() {
  std::vector<uint64_t> fa(32), fb(32);
  srand(7);
  for (int i = 0; i < 32; ++i) {
    fa[i] = ((uint64_t)rand() << 33) ^ ((uint64_t)rand() << 11) ^ rand();
    fb[i] = ((uint64_t)rand() << 35) ^ ((uint64_t)rand() << 13) ^ rand();
  }
  fa[3] = ~0ull; fb[5] = ~0ull; fa[0] = 0;
  std::vector<uint8_t> ea(32 * 32), eb(32 * 32);
  for (int i = 0; i < 32; ++i) { expand64(fa[i], &ea[i * 32]); expand64(fb[i], &eb[i * 32]); }
  uint8_t *dA, *dB; float* dD;
  hipMalloc(&dA, ea.size()); hipMalloc(&dB, eb.size()); hipMalloc(&dD, 32 * 32 * 4);
  hipMemcpy(dA, ea.data(), ea.size(), hipMemcpyHostToDevice);
  hipMemcpy(dB, eb.data(), eb.size(), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe<SCALE>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  std::vector<float> D(32 * 32);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0, badT = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      const float want = (float)__builtin_popcountll(fa[i] & fb[j]);
      if (D[i * 32 + j] != want) ++bad;
      if (D[j * 32 + i] != want) ++badT;
    }
  printf("scale=0x%08x: D[i][j]==popc(a_i&b_j) mismatches=%d (transposed reading mismatches=%d)  D[3][5]=%g D[0][0]=%g D[3][0]=%g want %d\n",
         SCALE, bad, badT, D[3 * 32 + 5], D[0], D[3 * 32], __builtin_popcountll(fa[3] & fb[0]));
  // rate
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  float* dO; hipMalloc(&dO, (size_t)p.multiProcessorCount * 2 * 256 * 4);
  for (int bpc : {1, 2}) {
    const int blocks = p.multiProcessorCount * bpc, iters = 2000;
    hipLaunchKernelGGL(rate<SCALE>, dim3(blocks), dim3(256), 0, 0, dO, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<SCALE>, dim3(blocks), dim3(256), 0, 0, dO, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * iters * 16;
    printf("   rate blocks/CU=%d: %.3f ms, %.1f TFLOP/s, %.2f T pair-bits/s -> %.2f T pairs/s at 2048 bits\n", bpc, ms,
           mfmas * 2 * 32 * 32 * 64 / (ms * 1e-3) / 1e12, mfmas * 32 * 32 * 64 / (ms * 1e-3) / 1e12,
           mfmas * 32 * 32 * 64 / (ms * 1e-3) / 2048 / 1e12);
  }
  return bad;
}

int main() {
  run<0x7f7f7f7f>();
  run<0>();
  return 0;
}
