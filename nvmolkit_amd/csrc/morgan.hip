// Batched Morgan fingerprints from flattened invariants — gfx950.
//
// Replaces morganFingerprintKernelBatch<maxAtoms, fpSize> / launchMorganFingerprintKernelBatch
// (reference: src/morgan_fingerprint_kernels.cu:151-485), i.e. RDKit's getEnvironments
// (src/morgan_fingerprint_cpu.cpp:61-255) with chirality off, bond types on, all atoms included, folded
// to fp_bits bits.  Inputs are the arrays of MorganInvariantsGenerator::ComputeInvariantsInto
// (src/morgan_fingerprint_common.cpp:43-124).
//
// Design for wave64 (the reference uses a cooperative-group tile of maxAtoms threads and a CUB merge sort
// of (neighbourhood, invariant, atom) tuples every round):
//   * lane = atom: ONE wave64 per molecule in the 64-atom bucket and TWO molecules per wave64 in the 32-atom bucket
//     (lanes 0-31 / 32-63, each half with its own LDS state: no idle half wave; the barrier is free in both),
//     two waves per molecule for the 128-atom bucket;
//   * no sort.  The reference's sorted sweep only decides, among atoms whose bond-neighbourhood bitsets are
//     EQUAL, which one comes first — the primary sort key is the bitset itself, so atoms with different
//     bitsets never interact.  An atom therefore survives a round iff its bitset was not accepted in an
//     earlier round and no other live atom has the same bitset with a smaller (invariant, atom index).
//     That is an all-pairs compare of <= 128 bitsets out of LDS, O(n) per lane, and needs no ordering of
//     bitsets at all (FlatBitVect::operator<, flat_bit_vect.h:219-237, is not needed);
//   * per-molecule state (invariants, neighbourhood bitsets, the list of accepted neighbourhoods, the folded
//     fingerprint) lives in LDS; the reference keeps the accepted list in a global scratch buffer;
//   * molecules of 256 to 1023 atoms (0.8 % of the reference's ChEMBL benchmark set; the reference computes them on the
//     CPU, src/morgan_fingerprint_gpu.cpp:181-188) run the same kernel with 512 / 1024 lanes and the three bitset arrays
//     in a per-workgroup global scratch (GSETS): 1024 bitsets of 1024 bits do not fit the LDS.
#include "common.h"

namespace nvmk {
namespace morgan {

constexpr int MAX_BONDS  = 8;  // kMaxBondsPerAtom / bondStride
constexpr int MAX_RADIUS = 8;

__device__ __forceinline__ void hash_combine(uint32_t& seed, const uint32_t v) {
  seed ^= v + 0x9e3779b9u + (seed << 6) + (seed >> 2);  // boost hash_combine on uint32 (kernels.cu:53-55)
}

template <int NW> struct Bits {
  uint32_t w[NW];
};

template <int NW> __device__ __forceinline__ bool bits_equal(const Bits<NW>& a, const uint32_t* b) {
  bool eq = true;
#pragma unroll
  for (int k = 0; k < NW; ++k) eq = eq && (a.w[k] == b[k]);
  return eq;
}

// NW = bitset words = stride / 32; BLOCK = threads per workgroup; MPB = molecules per workgroup (BLOCK / MPB >= stride
// lanes each).
template <int NW, int BLOCK, int MPB = 1, bool GSETS = false>
__global__ __launch_bounds__(BLOCK) void morgan_kernel(const uint32_t* __restrict__ atomInv,
                                                       const uint32_t* __restrict__ bondInv,
                                                       const int16_t* __restrict__ bondIdx,
                                                       const int16_t* __restrict__ bondOther,
                                                       const int16_t* __restrict__ nAtomsPerMol,
                                                       const int32_t* __restrict__ outIdx, const int64_t nMols,
                                                       const int radius, const int fpBits, uint32_t* __restrict__ out,
                                                       uint32_t* __restrict__ setScratch) {
  constexpr int STRIDE = NW * 32;
  constexpr int LPM    = BLOCK / MPB;  // lanes per molecule
  static_assert(LPM >= STRIDE, "one lane per atom slot");
  static_assert(!GSETS || MPB == 1, "the global-scratch form runs one molecule per workgroup");
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  const int    sub      = threadIdx.x / LPM;
  const size_t setWords = (2 + static_cast<size_t>(radius)) * STRIDE * NW;  // nbh, rnbh, seen
  const size_t molWords = 3 * STRIDE + (GSETS ? 0 : setWords) + fpBits / 32 + 4;
  char*        smem     = smem_all + sub * molWords * 4;
  uint32_t* cur      = reinterpret_cast<uint32_t*>(smem);        // [STRIDE] invariants entering this round
  uint32_t* rinv     = cur + STRIDE;                             // [STRIDE] invariants computed this round
  uint32_t* liveNow  = rinv + STRIDE;                            // [STRIDE] atom produced an environment this round
  uint32_t* nbh      = GSETS ? setScratch + static_cast<size_t>(blockIdx.x) * setWords : liveNow + STRIDE;  // [STRIDE][NW] previous round
  uint32_t* rnbh     = nbh + STRIDE * NW;                        // [STRIDE][NW] neighbourhoods of this round
  uint32_t* seen     = rnbh + STRIDE * NW;                       // [radius * STRIDE][NW] accepted neighbourhoods
  uint32_t* fp       = GSETS ? liveNow + STRIDE : seen + static_cast<size_t>(radius) * STRIDE * NW;  // [fpBits / 32]
  int*      seenCnt  = reinterpret_cast<int*>(fp + fpBits / 32);

  const int64_t molRaw = static_cast<int64_t>(blockIdx.x) * MPB + sub;
  const bool    valid  = molRaw < nMols;  // an odd tail leaves one half idle: it still takes part in the barriers
  const int64_t mol    = valid ? molRaw : 0;
  const int a      = threadIdx.x % LPM;
  const int n      = valid ? nAtomsPerMol[mol] : 0;
  const int words  = fpBits / 32;
  uint32_t* outRow = out + static_cast<int64_t>(outIdx ? outIdx[mol] : mol) * words;

  for (int k = a; k < words; k += LPM) fp[k] = 0u;
  if (a == 0) *seenCnt = 0;
  const bool     atom = a < n;
  const uint32_t inv0 = (atom && a < STRIDE) ? atomInv[mol * STRIDE + a] : 0u;

  // this lane's bonds
  int bIdx[MAX_BONDS], bOth[MAX_BONDS];
  int degree = 0;
  if (atom) {
    const int16_t* bi = bondIdx + (mol * STRIDE + a) * MAX_BONDS;
    const int16_t* bo = bondOther + (mol * STRIDE + a) * MAX_BONDS;
#pragma unroll
    for (int k = 0; k < MAX_BONDS; ++k) {
      bIdx[k] = bi[k];
      bOth[k] = bo[k];
    }
    // entries are packed from slot 0 (-1 padded), so the first -1 ends the list
#pragma unroll
    for (int k = 0; k < MAX_BONDS; ++k) {
      if (bIdx[k] >= 0 && degree == k) degree = k + 1;
    }
  }
  uint32_t btype[MAX_BONDS];
#pragma unroll
  for (int k = 0; k < MAX_BONDS; ++k) btype[k] = (k < degree) ? bondInv[mol * STRIDE + bIdx[k]] : 0u;

  if (a < STRIDE) {
    cur[a] = inv0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      nbh[a * NW + k]  = 0u;
      rnbh[a * NW + k] = 0u;
    }
  }
  __syncthreads();
  if (atom) atomicOr(&fp[(inv0 % static_cast<uint32_t>(fpBits)) >> 5], 1u << ((inv0 % static_cast<uint32_t>(fpBits)) & 31));

  bool     dead = !atom;
  Bits<NW> rn;
#pragma unroll
  for (int k = 0; k < NW; ++k) rn.w[k] = 0u;

  for (int layer = 0; layer < radius; ++layer) {
    bool     computed = false;
    uint32_t invar    = 0u;
    if (!dead) {
      if (degree == 0) {
        dead = true;  // isolated atom: never produces an environment (cpu.cpp:155-158)
      } else {
        // neighbourhood of this round + (bond type, neighbour invariant) pairs (cpu.cpp:167-181)
        int32_t  pt[MAX_BONDS];
        uint32_t pv[MAX_BONDS];
#pragma unroll
        for (int k = 0; k < MAX_BONDS; ++k) {
          if (k < degree) {
            const int b = bIdx[k], o = bOth[k];
#pragma unroll
            for (int w = 0; w < NW; ++w) {  // static register indices only (a dynamic index would spill rn)
              rn.w[w] |= nbh[o * NW + w] | (((b >> 5) == w) ? (1u << (b & 31)) : 0u);
            }
            pt[k] = static_cast<int32_t>(btype[k]);
            pv[k] = cur[o];
          } else {
            pt[k] = 0x7fffffff;  // sorts behind every real pair
            pv[k] = 0xffffffffu;
          }
        }
        // sort the <= 8 pairs ascending by (bond type, invariant): odd-even transposition, branch-free
#pragma unroll
        for (int pass = 0; pass < MAX_BONDS; ++pass) {
#pragma unroll
          for (int i = pass & 1; i + 1 < MAX_BONDS; i += 2) {
            const bool swap = (pt[i + 1] < pt[i]) || (pt[i + 1] == pt[i] && pv[i + 1] < pv[i]);
            const int32_t  t0 = swap ? pt[i + 1] : pt[i];
            const int32_t  t1 = swap ? pt[i] : pt[i + 1];
            const uint32_t v0 = swap ? pv[i + 1] : pv[i];
            const uint32_t v1 = swap ? pv[i] : pv[i + 1];
            pt[i] = t0; pt[i + 1] = t1; pv[i] = v0; pv[i + 1] = v1;
          }
        }
        invar = static_cast<uint32_t>(layer);  // cpu.cpp:187-194, pair hash kernels.cu:57-62
        hash_combine(invar, cur[a]);
#pragma unroll
        for (int k = 0; k < MAX_BONDS; ++k) {
          if (k < degree) {
            uint32_t ps = 0u;
            hash_combine(ps, static_cast<uint32_t>(pt[k]));
            hash_combine(ps, pv[k]);
            hash_combine(invar, ps);
          }
        }
        computed = true;
      }
    }
    if (a < STRIDE) {
      liveNow[a] = computed ? 1u : 0u;
      rinv[a]    = invar;
      if (computed) {
#pragma unroll
        for (int w = 0; w < NW; ++w) rnbh[a * NW + w] = rn.w[w];
      }
    }
    __syncthreads();

    // dedup without sorting (see header): lose to an earlier round's accepted copy, or to a live atom with
    // the same bitset and a smaller (invariant, index)
    bool accepted = false;
    if (computed) {
      bool      lose = false;
      const int ns   = *seenCnt;
      for (int s = 0; s < ns && !lose; ++s) lose = bits_equal<NW>(rn, seen + s * NW);
      for (int b = 0; b < n && !lose; ++b) {
        if (b != a && liveNow[b] != 0u && bits_equal<NW>(rn, rnbh + b * NW)) {
          const uint32_t vb = rinv[b];
          lose              = (vb < invar) || (vb == invar && b < a);
        }
      }
      accepted = !lose;
      dead     = lose;
    }
    __syncthreads();  // everyone has read `seen` before this round's winners are appended
    if (accepted) {
      const int slot = atomicAdd(seenCnt, 1);
#pragma unroll
      for (int w = 0; w < NW; ++w) seen[slot * NW + w] = rn.w[w];
      const uint32_t bit = invar % static_cast<uint32_t>(fpBits);
      atomicOr(&fp[bit >> 5], 1u << (bit & 31));
    }
    // roll: this round's ids become the invariants (0 where nothing was computed), neighbourhoods carry over
    if (a < STRIDE) {
      cur[a] = computed ? invar : 0u;
      if (computed) {
#pragma unroll
        for (int w = 0; w < NW; ++w) nbh[a * NW + w] = rn.w[w];
      }
    }
    __syncthreads();
  }
  __syncthreads();
  if (valid) {
    for (int k = a; k < words; k += LPM) outRow[k] = fp[k];
  }
}

template <int NW, int BLOCK, int MPB = 1, bool GSETS = false>
int launch_t(const uint32_t* atomInv, const uint32_t* bondInv, const int16_t* bondIdx, const int16_t* bondOther,
             const int16_t* nAtoms, const int32_t* outIdx, int64_t nMols, int radius, int fpBits, uint32_t* out,
             hipStream_t stream) {
  constexpr int STRIDE   = NW * 32;
  const size_t  setWords = (2 + static_cast<size_t>(radius)) * STRIDE * NW;
  const size_t  shmem    = MPB * (3 * STRIDE + (GSETS ? 0 : setWords) + fpBits / 32 + 4) * 4;
  auto          kern     = morgan_kernel<NW, BLOCK, MPB, GSETS>;
  if (shmem > 64 * 1024) {
    NVMK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(shmem)));
  }
  StreamScratch scratch;  // freed stream-ordered after the kernel
  if (GSETS) NVMK_HIP_CHECK(scratch.alloc(static_cast<size_t>(nMols) * setWords * 4, stream));
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>((nMols + MPB - 1) / MPB)), dim3(BLOCK), shmem, stream, atomInv, bondInv, bondIdx,
                     bondOther, nAtoms, outIdx, nMols, radius, fpBits, out, scratch.as<uint32_t>());
  NVMK_LAUNCH_CHECK();
  return NVMK_OK;
}

}  // namespace morgan
}  // namespace nvmk

extern "C" int nvmk_morgan_from_invariants(const uint32_t* d_atom_inv, const uint32_t* d_bond_inv,
                                           const int16_t* d_bond_idx, const int16_t* d_bond_other,
                                           const int16_t* d_n_atoms, const int32_t* d_out_idx, int64_t n_mols,
                                           int max_atoms, int radius, int fp_bits, uint32_t* d_out, void* stream) {
  NVMK_MARK_ENTRY();
  using namespace nvmk;
  NVMK_REQUIRE(max_atoms == 32 || max_atoms == 64 || max_atoms == 128 || max_atoms == 256 || max_atoms == 512 || max_atoms == 1024,
               "morgan: max_atoms must be 32, 64, 128, 256, 512 or 1024, got %d", max_atoms);
  NVMK_REQUIRE(radius >= 0 && radius <= morgan::MAX_RADIUS, "morgan: radius must be in [0, %d], got %d",
               morgan::MAX_RADIUS, radius);
  // reference: fpSize in {128, ..., 4096} (nvmolkit/fingerprints.cpp:66-90 -> std::invalid_argument otherwise)
  NVMK_REQUIRE(fp_bits == 128 || fp_bits == 256 || fp_bits == 512 || fp_bits == 1024 || fp_bits == 2048 || fp_bits == 4096,
               "Unsupported fpSize %d: must be one of 128, 256, 512, 1024, 2048, 4096", fp_bits);
  NVMK_REQUIRE(n_mols >= 0 && n_mols <= 0x7fffffffLL, "morgan: bad molecule count %lld", (long long)n_mols);
  if (n_mols == 0) return NVMK_OK;
  NVMK_REQUIRE(d_atom_inv && d_bond_inv && d_bond_idx && d_bond_other && d_n_atoms && d_out, "morgan: NULL buffer");
  hipStream_t s = as_stream(stream);
  switch (max_atoms) {
    case 32:
      return morgan::launch_t<1, 64, 2>(d_atom_inv, d_bond_inv, d_bond_idx, d_bond_other, d_n_atoms, d_out_idx, n_mols, radius,
                                     fp_bits, d_out, s);
    case 64:
      return morgan::launch_t<2, 64>(d_atom_inv, d_bond_inv, d_bond_idx, d_bond_other, d_n_atoms, d_out_idx, n_mols, radius,
                                     fp_bits, d_out, s);
    case 128:
      return morgan::launch_t<4, 128>(d_atom_inv, d_bond_inv, d_bond_idx, d_bond_other, d_n_atoms, d_out_idx, n_mols, radius,
                                      fp_bits, d_out, s);
    case 256:
      return morgan::launch_t<8, 256>(d_atom_inv, d_bond_inv, d_bond_idx, d_bond_other, d_n_atoms, d_out_idx, n_mols, radius,
                                      fp_bits, d_out, s);
    case 512:
      return morgan::launch_t<16, 512, 1, true>(d_atom_inv, d_bond_inv, d_bond_idx, d_bond_other, d_n_atoms, d_out_idx, n_mols,
                                                radius, fp_bits, d_out, s);
    default:
      return morgan::launch_t<32, 1024, 1, true>(d_atom_inv, d_bond_inv, d_bond_idx, d_bond_other, d_n_atoms, d_out_idx, n_mols,
                                                 radius, fp_bits, d_out, s);
  }
}
