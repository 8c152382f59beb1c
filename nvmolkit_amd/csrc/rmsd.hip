// Pairwise conformer RMSD (optionally after optimal superposition) and greedy RMS pruning — gfx950.
//
// Replaces (reference paths):
//   src/conformer_rmsd.cu:30-392          conformerRmsdMatrixGpu / conformerRmsdBatchMatrixGpu (one 128-thread block per
//                                         pair, 11 cub block reductions, Cardano eigenvalues on thread 0)
//   rdkit_extensions/conformer_pruning.cpp:88-137   _isConfFarFromRest / addConformersToMoleculeWithPruning (CPU loop)
//
// MI355X-first: a conformer pair is ONE wave64 (atoms strided over lanes, the 17 partial sums reduced with wave shuffles,
// no LDS and no barrier), four pairs per workgroup; molecules of a batch are flattened into one launch through a pair
// offset table.  RMSD^2 = (Sp + Sq - 2 (s0 + s1 + sgn(det H) s2)) / N with s_k the singular values of the 3x3
// cross-covariance H of the centred coordinates (Kabsch, proper rotations only), s_k^2 = eigenvalues of H^T H from
// the trigonometric solution of the characteristic cubic.  Pruning keeps conformer i iff its RMSD to every conformer
// kept before it is >= the threshold (the reference compares sums of squares: ssr < n thr^2 rejects).
#include <cmath>

#include "common.h"

namespace nvmk {
namespace rmsd {

constexpr int NT = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// eigenvalues of a symmetric 3x3 matrix, descending
__device__ __forceinline__ void sym_eig3(const double a00, const double a01, const double a02, const double a11, const double a12,
                                         const double a22, double& e0, double& e1, double& e2) {
  const double p  = a00 + a11 + a22;
  const double q  = a00 * a11 + a00 * a22 + a11 * a22 - a01 * a01 - a02 * a02 - a12 * a12;
  const double r  = a00 * a11 * a22 + 2.0 * a01 * a02 * a12 - a00 * a12 * a12 - a11 * a02 * a02 - a22 * a01 * a01;
  const double p3 = p / 3.0;
  const double pp = (p * p - 3.0 * q) / 9.0;
  const double qq = (2.0 * p * p * p - 9.0 * p * q + 27.0 * r) / 54.0;
  const double sp = sqrt(fmax(pp, 0.0));
  const double th = acos(fmin(fmax(qq / fmax(sp * sp * sp, 1.0e-30), -1.0), 1.0)) / 3.0;
  constexpr double kTwoPiOver3 = 2.0943951023931954923;
  double x0 = 2.0 * sp * cos(th) + p3, x1 = 2.0 * sp * cos(th - kTwoPiOver3) + p3, x2 = 2.0 * sp * cos(th - 2.0 * kTwoPiOver3) + p3;
  if (x1 > x0) { const double t = x0; x0 = x1; x1 = t; }
  if (x2 > x0) { const double t = x0; x0 = x2; x2 = t; }
  if (x2 > x1) { const double t = x1; x1 = x2; x2 = t; }
  e0 = x0;
  e1 = x1;
  e2 = x2;
}

// RMSD after optimal superposition from the centred sums: Sp = sum |p|^2, Sq = sum |q|^2, H = sum p q^T, invN = 1 / points.
__device__ __forceinline__ double kabsch_rmsd(const double sp, const double sq, const double (&H)[9], const double invN) {
  const double g00 = H[0] * H[0] + H[3] * H[3] + H[6] * H[6], g01 = H[0] * H[1] + H[3] * H[4] + H[6] * H[7];
  const double g02 = H[0] * H[2] + H[3] * H[5] + H[6] * H[8], g11 = H[1] * H[1] + H[4] * H[4] + H[7] * H[7];
  const double g12 = H[1] * H[2] + H[4] * H[5] + H[7] * H[8], g22 = H[2] * H[2] + H[5] * H[5] + H[8] * H[8];
  double       e0, e1, e2;
  sym_eig3(g00, g01, g02, g11, g12, g22, e0, e1, e2);
  const double det = H[0] * (H[4] * H[8] - H[5] * H[7]) - H[1] * (H[3] * H[8] - H[5] * H[6]) + H[2] * (H[3] * H[7] - H[4] * H[6]);
  // The trigonometric roots lose the small eigenvalues of a (nearly) rank-deficient H — 2 or 3 atoms, planar
  // molecules — to cancellation (1e-12 absolute, i.e. 1e-6 in the singular value).  Keep the largest root and deflate:
  // e1 + e2 = trace - e0, e1 e2 = det(G) / e0 = det(H)^2 / e0, smaller root from the product.
  if (e0 > 0.0) {
    {  // two Newton steps on the characteristic polynomial polish the (well separated) largest root
      const double p = g00 + g11 + g22;
      const double q = g00 * g11 + g00 * g22 + g11 * g22 - g01 * g01 - g02 * g02 - g12 * g12;
      const double r = det * det;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const double f  = ((e0 - p) * e0 + q) * e0 - r;
        const double fp = (3.0 * e0 - 2.0 * p) * e0 + q;
        if (fp > 0.0) e0 = fmin(fmax(e0 - f / fp, 0.0), p);
      }
    }
    const double sum  = fmax(g00 + g11 + g22 - e0, 0.0);
    const double prod = det * det / e0;
    const double disc = sqrt(fmax(sum * sum - 4.0 * prod, 0.0));
    e1                = 0.5 * (sum + disc);
    e2                = e1 > 0.0 ? fmin(prod / e1, e1) : 0.0;
  }
  const double s0 = sqrt(fmax(e0, 0.0)), s1 = sqrt(fmax(e1, 0.0));
  double       s2 = sqrt(fmax(e2, 0.0));
  if (det < 0.0) s2 = -s2;  // the best PROPER rotation: flip the smallest singular value
  return sqrt(fmax((sp + sq - 2.0 * (s0 + s1 + s2)) * invN, 0.0));
}

// One wave per pair.  pairOffsets[m] = first pair of molecule m; pair q of a molecule is (i, j), i > j, q = i (i - 1) / 2 + j.
__global__ __launch_bounds__(NT) void rmsd_pairs_kernel(const double* __restrict__ coords, const int64_t* __restrict__ coordOffsets,
                                                        const int32_t* __restrict__ nAtoms, const int64_t* __restrict__ pairOffsets,
                                                        const int nMols, const int64_t totalPairs, const int prealigned,
                                                        double* __restrict__ out) {
  const int     lane = threadIdx.x & 63;
  const int64_t pair = static_cast<int64_t>(blockIdx.x) * (NT / 64) + (threadIdx.x >> 6);
  if (pair >= totalPairs) return;
  // molecule of this pair: last m with pairOffsets[m] <= pair
  int lo = 0, hi = nMols - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (pairOffsets[mid] <= pair) {
      lo = mid;
    } else {
      hi = mid - 1;
    }
  }
  const int     m = lo;
  const int64_t q = pair - pairOffsets[m];
  int64_t       i = static_cast<int64_t>((1.0 + sqrt(1.0 + 8.0 * static_cast<double>(q))) * 0.5);
  while (i * (i - 1) / 2 > q) --i;  // guard the floating-point estimate
  while ((i + 1) * i / 2 <= q) ++i;
  const int64_t j = q - i * (i - 1) / 2;
  const int     n = nAtoms[m];
  const double* A = coords + coordOffsets[m] + i * n * 3;
  const double* B = coords + coordOffsets[m] + j * n * 3;
  const double  invN = 1.0 / static_cast<double>(n);

  if (prealigned) {  // raw coordinates, no centring (RDKit's prealigned=True)
    double s = 0.0;
    for (int a = lane; a < n; a += 64) {
      const double dx = A[3 * a] - B[3 * a], dy = A[3 * a + 1] - B[3 * a + 1], dz = A[3 * a + 2] - B[3 * a + 2];
      s += dx * dx + dy * dy + dz * dz;
    }
    s = wave_sum(s);
    if (lane == 0) out[pair] = sqrt(s * invN);
    return;
  }
  double cA[3] = {0, 0, 0}, cB[3] = {0, 0, 0};
  for (int a = lane; a < n; a += 64) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      cA[c] += A[3 * a + c];
      cB[c] += B[3 * a + c];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    cA[c] = wave_sum(cA[c]) * invN;
    cB[c] = wave_sum(cB[c]) * invN;
  }
  double sp = 0.0, sq = 0.0, H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int a = lane; a < n; a += 64) {
    double p[3], r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      p[c] = A[3 * a + c] - cA[c];
      r[c] = B[3 * a + c] - cB[c];
    }
    sp += p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    sq += r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
#pragma unroll
      for (int y = 0; y < 3; ++y) H[3 * x + y] += p[x] * r[y];
    }
  }
  sp = wave_sum(sp);
  sq = wave_sum(sq);
#pragma unroll
  for (int k = 0; k < 9; ++k) H[k] = wave_sum(H[k]);
  if (lane == 0) out[pair] = kabsch_rmsd(sp, sq, H, invN);
}
// Symmetry-aware form: one wave per pair, the molecule's mappings one after the other; reference points = conformer i's atoms
// of mapping 0, probe points = conformer j's atoms of mapping k; the pair's entry is the smallest RMSD over k.
__global__ __launch_bounds__(NT) void rmsd_pairs_sym_kernel(const double* __restrict__ coords, const int64_t* __restrict__ coordOffsets,
                                                            const int32_t* __restrict__ nAtoms, const int64_t* __restrict__ pairOffsets,
                                                            const int nMols, const int64_t totalPairs,
                                                            const int64_t* __restrict__ matchOffsets, const int32_t* __restrict__ matchLen,
                                                            const int32_t* __restrict__ matches, double* __restrict__ out) {
  const int     lane = threadIdx.x & 63;
  const int64_t pair = static_cast<int64_t>(blockIdx.x) * (NT / 64) + (threadIdx.x >> 6);
  if (pair >= totalPairs) return;
  int lo = 0, hi = nMols - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (pairOffsets[mid] <= pair) {
      lo = mid;
    } else {
      hi = mid - 1;
    }
  }
  const int     m = lo;
  const int64_t q = pair - pairOffsets[m];
  int64_t       i = static_cast<int64_t>((1.0 + sqrt(1.0 + 8.0 * static_cast<double>(q))) * 0.5);
  while (i * (i - 1) / 2 > q) --i;
  while ((i + 1) * i / 2 <= q) ++i;
  const int64_t  j  = q - i * (i - 1) / 2;
  const int      n  = nAtoms[m], L = matchLen[m];
  const int      K  = L > 0 ? static_cast<int>((matchOffsets[m + 1] - matchOffsets[m]) / L) : 0;
  const double*  A  = coords + coordOffsets[m] + i * n * 3;
  const double*  B  = coords + coordOffsets[m] + j * n * 3;
  const int32_t* M0 = matches + matchOffsets[m];
  const double   invL = L > 0 ? 1.0 / static_cast<double>(L) : 0.0;
  double cA[3] = {0, 0, 0}, sp = 0.0;
  for (int a = lane; a < L; a += 64) {
#pragma unroll
    for (int c = 0; c < 3; ++c) cA[c] += A[3 * M0[a] + c];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) cA[c] = wave_sum(cA[c]) * invL;
  for (int a = lane; a < L; a += 64) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double p = A[3 * M0[a] + c] - cA[c];
      sp += p * p;
    }
  }
  sp = wave_sum(sp);
  double best = 1.0e300;
  for (int k = 0; k < K; ++k) {
    const int32_t* Mk = M0 + static_cast<int64_t>(k) * L;
    double         cB[3] = {0, 0, 0};
    for (int a = lane; a < L; a += 64) {
#pragma unroll
      for (int c = 0; c < 3; ++c) cB[c] += B[3 * Mk[a] + c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) cB[c] = wave_sum(cB[c]) * invL;
    double sq = 0.0, H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = lane; a < L; a += 64) {
      double p[3], r[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        p[c] = A[3 * M0[a] + c] - cA[c];
        r[c] = B[3 * Mk[a] + c] - cB[c];
      }
      sq += r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
#pragma unroll
      for (int x = 0; x < 3; ++x) {
#pragma unroll
        for (int y = 0; y < 3; ++y) H[3 * x + y] += p[x] * r[y];
      }
    }
    sq = wave_sum(sq);
#pragma unroll
    for (int t = 0; t < 9; ++t) H[t] = wave_sum(H[t]);
    best = fmin(best, kabsch_rmsd(sp, sq, H, invL));  // (every lane computes it: no divergence, no broadcast)
  }
  if (lane == 0) out[pair] = K > 0 ? best : 0.0;
}

// Greedy pruning, one wave per molecule: conformer i is kept iff rmsd(i, k) >= thr for every kept k < i.
__global__ __launch_bounds__(64) void prune_kernel(const double* __restrict__ rmsd, const int64_t* __restrict__ pairOffsets,
                                                   const int32_t* __restrict__ confStarts, const int nMols, const double thr,
                                                   uint8_t* __restrict__ keep) {
  const int m = blockIdx.x;
  if (m >= nMols) return;
  const int      lane = threadIdx.x;
  const int      c0 = confStarts[m], n = confStarts[m + 1] - c0;
  const double*  R  = rmsd + pairOffsets[m];
  volatile uint8_t* K = keep + c0;  // written by lane 0, read by every lane in later iterations: bypass the L1
  for (int i = 0; i < n; ++i) {
    bool close = false;
    for (int k = lane; k < i; k += 64) {
      if (K[k] && R[static_cast<int64_t>(i) * (i - 1) / 2 + k] < thr) close = true;
    }
    const bool any = __ballot(close) != 0ull;
    if (lane == 0) K[i] = any ? 0 : 1;
    __threadfence();
  }
}

}  // namespace rmsd
}  // namespace nvmk

using namespace nvmk;

extern "C" {

int nvmk_conformer_rmsd_batch(const double* d_coords, const int64_t* d_coord_offsets, const int32_t* d_n_atoms,
                              const int64_t* d_pair_offsets, int n_mols, int64_t total_pairs, int prealigned, double* d_out,
                              void* stream) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(n_mols >= 0 && total_pairs >= 0, "conformer rmsd: negative size");
  if (n_mols == 0 || total_pairs == 0) return NVMK_OK;
  NVMK_REQUIRE(d_coords && d_coord_offsets && d_n_atoms && d_pair_offsets && d_out, "conformer rmsd: NULL buffer");
  const int64_t blocks = ceil_div<int64_t>(total_pairs, rmsd::NT / 64);
  NVMK_REQUIRE(blocks <= 0x7fffffffLL, "conformer rmsd: too many pairs (%lld)", (long long)total_pairs);
  hipLaunchKernelGGL(rmsd::rmsd_pairs_kernel, dim3(static_cast<unsigned>(blocks)), dim3(rmsd::NT), 0, as_stream(stream), d_coords,
                     d_coord_offsets, d_n_atoms, d_pair_offsets, n_mols, total_pairs, prealigned, d_out);
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

int nvmk_conformer_rmsd_batch_sym(const double* d_coords, const int64_t* d_coord_offsets, const int32_t* d_n_atoms,
                                  const int64_t* d_pair_offsets, int n_mols, int64_t total_pairs, const int64_t* d_match_offsets,
                                  const int32_t* d_match_len, const int32_t* d_matches, double* d_out, void* stream) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(n_mols >= 0 && total_pairs >= 0, "conformer rmsd: negative size");
  if (n_mols == 0 || total_pairs == 0) return NVMK_OK;
  NVMK_REQUIRE(d_coords && d_coord_offsets && d_n_atoms && d_pair_offsets && d_out, "conformer rmsd: NULL buffer");
  NVMK_REQUIRE(d_match_offsets && d_match_len && d_matches, "conformer rmsd: NULL match table");
  const int64_t blocks = ceil_div<int64_t>(total_pairs, rmsd::NT / 64);
  NVMK_REQUIRE(blocks <= 0x7fffffffLL, "conformer rmsd: too many pairs (%lld)", (long long)total_pairs);
  hipLaunchKernelGGL(rmsd::rmsd_pairs_sym_kernel, dim3(static_cast<unsigned>(blocks)), dim3(rmsd::NT), 0, as_stream(stream), d_coords,
                     d_coord_offsets, d_n_atoms, d_pair_offsets, n_mols, total_pairs, d_match_offsets, d_match_len, d_matches, d_out);
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

int nvmk_conformer_prune(const double* d_rmsd, const int64_t* d_pair_offsets, const int32_t* d_conf_starts, int n_mols,
                         double threshold, uint8_t* d_keep, void* stream) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(n_mols >= 0, "conformer prune: negative size");
  NVMK_REQUIRE(threshold >= 0.0, "conformer prune: negative threshold");
  if (n_mols == 0) return NVMK_OK;
  NVMK_REQUIRE(d_pair_offsets && d_conf_starts && d_keep, "conformer prune: NULL buffer");
  hipLaunchKernelGGL(rmsd::prune_kernel, dim3(static_cast<unsigned>(n_mols)), dim3(64), 0, as_stream(stream), d_rmsd, d_pair_offsets,
                     d_conf_starts, n_mols, threshold, d_keep);
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

}  // extern "C"
