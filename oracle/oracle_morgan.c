/*
 * oracle_morgan.c — CPU restatement of the reference's Morgan fingerprint path on a FLATTENED graph.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_similarity.c header).
 *
 * Follows getEnvironments / computeFpFromEnvironments (src/morgan_fingerprint_cpu.cpp:61-255, :257-275,
 * itself "adapted from RDKit environment code") with includeChirality = false, onlyNonzeroInvariants =
 * false and all atoms included (getFingerprintImpl, :279-305), and the 32-bit hash arithmetic spelled out
 * in the reference kernel (src/morgan_fingerprint_kernels.cu:53-62: boost hash_combine on uint32, a pair
 * is hashed as combine(combine(0, first), second)).  The graph comes in the layout produced by
 * MorganInvariantsGenerator::ComputeInvariantsInto (src/morgan_fingerprint_common.cpp:43-124):
 * per molecule `stride` atom invariants, `stride` bond invariants (indexed by bond id) and per atom up to 8
 * (bond id, other atom) pairs, -1 padded.
 *
 * Pinning status: PINNED to RDKit through the values RDKit itself publishes — the environment-deduplication logic through
 * the element-count known answers RDKit's own test-suite holds and the reference repeats
 * (tests/test_morgan_fingerprint_ref.cpp:44-60) on hand-flattened graphs (tests/test_oracle_morgan.py); the invariant
 * recipe and the hash chain through the identifiers of RDKit's documentation (the 16 environments of c1cccnc1C with
 * 98513984 twice at radius 1 and 4048591891 at radius 2, bit 872 of c1ccccc1CC1CC1, the radius-0 identifiers of CH3 / CH2
 * / OH / NH2 / aromatic CH, benzene's and ethanol's count fingerprints: tests/test_morgan_rdkit_known_answers.py, reached
 * through the library's SMILES ingestion); the hash arithmetic also by hand.  Not available in the project's images: a
 * corpus-wide comparison with RDKit's own fingerprints (tests/test_against_rdkit.py runs it wherever RDKit is installed).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MORGAN_MAX_BONDS 8 /* kMaxBondsPerAtom / bondStride */

static inline void hash_combine_u32(uint32_t* seed, uint32_t v) {
  *seed ^= v + 0x9e3779b9u + (*seed << 6) + (*seed >> 2);
}

/* gboost::hash<std::vector<uint32_t>> = hash_range with a 32-bit seed
 * (src/morgan_fingerprint_common.cpp:54,121; width inferred from the reference kernel). */
uint32_t orc_morgan_hash_vector(const uint32_t* v, int n) {
  uint32_t seed = 0;
  for (int i = 0; i < n; ++i) hash_combine_u32(&seed, v[i]);
  return seed;
}

typedef struct {
  int32_t  bond_type;
  uint32_t inv;
} nbr_pair;

static int pair_less(const nbr_pair* a, const nbr_pair* b) {
  if (a->bond_type != b->bond_type) return a->bond_type < b->bond_type;
  return a->inv < b->inv;
}

/* boost::dynamic_bitset / FlatBitVect ordering: most significant word first
 * (src/data_structures/flat_bit_vect.h:219-237). */
static int bits_cmp(const uint32_t* a, const uint32_t* b, int nw) {
  for (int w = nw - 1; w >= 0; --w) {
    if (a[w] != b[w]) return a[w] < b[w] ? -1 : 1;
  }
  return 0;
}

typedef struct {
  const uint32_t* bits;
  uint32_t        invar;
  uint32_t        atom;
} accum;

static int g_nw; /* qsort context (single-threaded use per call) */
static int accum_cmp(const void* pa, const void* pb) {
  const accum* a = (const accum*)pa;
  const accum* b = (const accum*)pb;
  const int    c = bits_cmp(a->bits, b->bits, g_nw);
  if (c != 0) return c;
  if (a->invar != b->invar) return a->invar < b->invar ? -1 : 1;
  if (a->atom != b->atom) return a->atom < b->atom ? -1 : 1;
  return 0;
}

/*
 * Environments of ONE molecule.  Writes up to (radius + 1) * n_atoms codes (and their layers) and returns
 * how many were produced.  n_bits_words = words of the bond bitsets (>= ceil(n_bonds / 32)).
 */
int orc_morgan_environments(const uint32_t* atom_inv, const uint32_t* bond_inv, const int16_t* bond_idx,
                            const int16_t* bond_other, int n_atoms, int radius, uint32_t* codes_out,
                            int32_t* layers_out) {
  if (n_atoms <= 0) return 0;
  int max_bond = -1;
  for (int a = 0; a < n_atoms; ++a)
    for (int k = 0; k < MORGAN_MAX_BONDS; ++k) {
      const int b = bond_idx[a * MORGAN_MAX_BONDS + k];
      if (b < 0) break;
      if (b > max_bond) max_bond = b;
    }
  const int nw = (max_bond + 1 + 31) / 32 > 0 ? (max_bond + 1 + 31) / 32 : 1;
  g_nw         = nw;

  uint32_t* cur       = (uint32_t*)calloc((size_t)n_atoms, sizeof(uint32_t));
  uint32_t* next      = (uint32_t*)calloc((size_t)n_atoms, sizeof(uint32_t));
  uint32_t* nbh       = (uint32_t*)calloc((size_t)n_atoms * nw, sizeof(uint32_t));  /* atomNeighborhoods */
  uint32_t* rnbh      = (uint32_t*)calloc((size_t)n_atoms * nw, sizeof(uint32_t));  /* roundAtomNeighborhoods */
  uint8_t*  dead      = (uint8_t*)calloc((size_t)n_atoms, 1);
  uint32_t* seen      = (uint32_t*)calloc((size_t)n_atoms * (radius + 1) * nw, sizeof(uint32_t));
  int       n_seen    = 0;
  accum*    round_acc = (accum*)malloc(sizeof(accum) * (size_t)n_atoms);
  int       n_out     = 0;
  memcpy(cur, atom_inv, sizeof(uint32_t) * (size_t)n_atoms);

  /* round 0: every atom's invariant (:136-143) */
  for (int a = 0; a < n_atoms; ++a) {
    codes_out[n_out]  = cur[a];
    layers_out[n_out] = 0;
    ++n_out;
  }

  for (int layer = 0; layer < radius; ++layer) {
    int n_acc = 0;
    for (int a = 0; a < n_atoms; ++a) { /* atomOrder is the identity (:129-133) */
      if (dead[a]) continue;
      const int16_t* bi = bond_idx + a * MORGAN_MAX_BONDS;
      const int16_t* bo = bond_other + a * MORGAN_MAX_BONDS;
      if (bi[0] < 0) { /* degree 0 (:155-158) */
        dead[a] = 1;
        continue;
      }
      nbr_pair pairs[MORGAN_MAX_BONDS];
      int      np = 0;
      uint32_t* rn = rnbh + (size_t)a * nw;
      for (int k = 0; k < MORGAN_MAX_BONDS && bi[k] >= 0; ++k) { /* :167-181 */
        const int b = bi[k], o = bo[k];
        rn[b / 32] |= 1u << (b % 32);
        for (int w = 0; w < nw; ++w) rn[w] |= nbh[(size_t)o * nw + w];
        pairs[np].bond_type = (int32_t)bond_inv[b];
        pairs[np].inv       = cur[o];
        ++np;
      }
      for (int i = 1; i < np; ++i) { /* std::sort of the pairs (:184) */
        nbr_pair key = pairs[i];
        int      j   = i - 1;
        while (j >= 0 && pair_less(&key, &pairs[j])) {
          pairs[j + 1] = pairs[j];
          --j;
        }
        pairs[j + 1] = key;
      }
      uint32_t invar = (uint32_t)layer; /* :187-189 */
      hash_combine_u32(&invar, cur[a]);
      for (int i = 0; i < np; ++i) { /* :192-194, pair hash per src/morgan_fingerprint_kernels.cu:57-62 */
        uint32_t ps = 0;
        hash_combine_u32(&ps, (uint32_t)pairs[i].bond_type);
        hash_combine_u32(&ps, pairs[i].inv);
        hash_combine_u32(&invar, ps);
      }
      next[a]                = invar; /* :222 */
      round_acc[n_acc].bits  = rn;    /* :226-228 */
      round_acc[n_acc].invar = invar;
      round_acc[n_acc].atom  = (uint32_t)a;
      ++n_acc;
    }
    qsort(round_acc, (size_t)n_acc, sizeof(accum), accum_cmp); /* :232 */
    for (int i = 0; i < n_acc; ++i) {                           /* :234-249 */
      int found = 0;
      for (int s = 0; s < n_seen && !found; ++s) found = bits_cmp(seen + (size_t)s * nw, round_acc[i].bits, nw) == 0;
      if (!found) {
        codes_out[n_out]  = round_acc[i].invar;
        layers_out[n_out] = layer + 1;
        ++n_out;
        memcpy(seen + (size_t)n_seen * nw, round_acc[i].bits, sizeof(uint32_t) * (size_t)nw);
        ++n_seen;
      } else {
        dead[round_acc[i].atom] = 1;
      }
    }
    /* :252-257: this round's ids become the invariants, untouched slots read 0; neighbourhoods roll */
    memcpy(cur, next, sizeof(uint32_t) * (size_t)n_atoms);
    memset(next, 0, sizeof(uint32_t) * (size_t)n_atoms);
    memcpy(nbh, rnbh, sizeof(uint32_t) * (size_t)n_atoms * nw);
  }
  free(cur);
  free(next);
  free(nbh);
  free(rnbh);
  free(dead);
  free(seen);
  free(round_acc);
  return n_out;
}

/*
 * Batch of molecules in the ComputeInvariantsInto layout -> folded bit vectors
 * (computeFpFromEnvironments :257-275 + getFingerprintImpl :298-302: bit = code % fpSize, set once).
 * out: n_mols rows of fp_bits/32 words, bit j = bit j%32 of word j/32 (FlatBitVect, flat_bit_vect.h).
 */
void orc_morgan_fingerprints(const uint32_t* atom_inv, const uint32_t* bond_inv, const int16_t* bond_idx,
                             const int16_t* bond_other, const int16_t* n_atoms, int64_t n_mols, int stride,
                             int radius, int fp_bits, uint32_t* out) {
  const int words = fp_bits / 32;
  uint32_t* codes  = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)stride * (size_t)(radius + 1));
  int32_t*  layers = (int32_t*)malloc(sizeof(int32_t) * (size_t)stride * (size_t)(radius + 1));
  for (int64_t m = 0; m < n_mols; ++m) {
    uint32_t* row = out + m * words;
    memset(row, 0, sizeof(uint32_t) * (size_t)words);
    const int n = orc_morgan_environments(atom_inv + m * stride, bond_inv + m * stride,
                                          bond_idx + m * stride * MORGAN_MAX_BONDS,
                                          bond_other + m * stride * MORGAN_MAX_BONDS, n_atoms[m], radius, codes, layers);
    for (int i = 0; i < n; ++i) {
      const uint32_t bit = codes[i] % (uint32_t)fp_bits;
      row[bit / 32] |= 1u << (bit % 32);
    }
  }
  free(codes);
  free(layers);
}
