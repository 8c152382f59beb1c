// Store-pattern microbenchmark: how fast can a 128x128-double tile grid be written to HBM with the
// lane->address patterns available to the dense similarity kernel's epilogue?  (tools only; not shipped)
//   P0  8-B nontemporal stores, MFMA C layout (2 x 256-B row segments per wave instruction)   [current kernel]
//   P1  as P0 with plain stores
//   P2  16-B nontemporal stores, 4 x 256-B row segments per wave instruction (lane-pair exchange layout)
//   P3  16-B nontemporal stores, one full 1-KB tile row per wave instruction (LDS-transposed layout)
//   P4  as P3 with plain stores
//   P5  as P3, 512 B segments (2 rows x 512 B)
//   P6 / P7 / P8  as P0 with cache-policy bits sc0 sc1 / sc0 sc1 nt / sc1 nt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int SUPER = 64;
__device__ inline void tile_of(int& tr, int& tc, int tilesR, int tilesC) {
  // same supertile walk as the real kernel: 64 x 64 tiles per supertile, row-major inside
  const long b = blockIdx.x;
  const int superCols = (tilesC + SUPER - 1) / SUPER;
  const long perSuper = (long)SUPER * SUPER;
  const long s = b / perSuper; const int w = (int)(b % perSuper);
  const int sr = (int)(s / superCols), sc = (int)(s % superCols);
  tr = sr * SUPER + w / SUPER; tc = sc * SUPER + w % SUPER;
}

template <int P> __global__ __launch_bounds__(256) void store_kernel(double* out, long ld, int tilesR, int tilesC) {
  int tr, tc; tile_of(tr, tc, tilesR, tilesC);
  if (tr >= tilesR || tc >= tilesC) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* base = out + (long)tr * 128 * ld + (long)tc * 128;
  const double v = (double)(tr * 131 + tc + lane);
  if (P == 0 || P == 1) {
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wr + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          double* p = base + (long)row * ld + wc + j * 32 + (lane & 31);
          if (P == 0) __builtin_nontemporal_store(v + r, p); else *p = v + r;
        }
  } else if (P == 2) {
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int row = wr + i * 32 + r * 4 + (lane >> 4);
          double2* p = reinterpret_cast<double2*>(base + (long)row * ld + wc + j * 32 + (lane & 15) * 2);
          double2 x; x.x = v + r; x.y = v - r;
          __builtin_nontemporal_store(x.x, &p->x); __builtin_nontemporal_store(x.y, &p->y);
        }
  } else if (P == 3 || P == 4) {
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const int row = wave * 32 + r;
      double* p = base + (long)row * ld + lane * 2;
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2 x; x.x = v + r; x.y = v - r;
      if (P == 3) __builtin_nontemporal_store(x, reinterpret_cast<d2*>(p)); else *reinterpret_cast<d2*>(p) = x;
    }
  } else if (P == 6 || P == 7 || P == 8) {
    // cache-policy variants of P0 (8-B stores in the MFMA C layout) via the instruction's sc0 / sc1 / nt bits
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wr + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          double*   p   = base + (long)row * ld + wc + j * 32 + (lane & 31);
          const double x = v + r;
          if (P == 6) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(x) : "memory");
          if (P == 7) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(x) : "memory");
          if (P == 8) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(p), "v"(x) : "memory");
        }
  } else if (P == 5) {
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const int row = wave * 32 + (r >> 1) * 2 + (lane >> 5), half = r & 1;
      double* p = base + (long)row * ld + half * 64 + (lane & 31) * 2;
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2 x; x.x = v + r; x.y = v - r;
      __builtin_nontemporal_store(x, reinterpret_cast<d2*>(p));
    }
  }
}

template <int P> void run(double* out, long ld, int tilesR, int tilesC) {
  const int superCols = (tilesC + SUPER - 1) / SUPER, superRows = (tilesR + SUPER - 1) / SUPER;
  const long blocks = (long)superCols * superRows * SUPER * SUPER;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  store_kernel<P><<<dim3((unsigned)blocks), 256>>>(out, ld, tilesR, tilesC);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < 3; ++i) store_kernel<P><<<dim3((unsigned)blocks), 256>>>(out, ld, tilesR, tilesC);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
  const double bytes = (double)tilesR * tilesC * 128 * 128 * 8;
  printf("P%d  %.3f ms  %.2f TB/s\n", P, ms, bytes / ms / 1e9);
}

int main() {
  const int tilesR = 64, tilesC = 7813;
  const long ld = (long)tilesC * 128;
  double* out; CK(hipMalloc(&out, (size_t)tilesR * 128 * ld * 8));
  run<0>(out, ld, tilesR, tilesC); run<1>(out, ld, tilesR, tilesC); run<2>(out, ld, tilesR, tilesC);
  run<3>(out, ld, tilesR, tilesC); run<4>(out, ld, tilesR, tilesC); run<5>(out, ld, tilesR, tilesC);
  run<6>(out, ld, tilesR, tilesC); run<7>(out, ld, tilesR, tilesC); run<8>(out, ld, tilesR, tilesC);
  CK(hipMemset(out, 0, (size_t)tilesR * 128 * ld * 8)); CK(hipDeviceSynchronize());
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a)); CK(hipMemsetAsync(out, 1, (size_t)tilesR * 128 * ld * 8)); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("hipMemset %.3f ms %.2f TB/s\n", ms, (double)tilesR * 128 * ld * 8 / ms / 1e9);
  return 0;
}
