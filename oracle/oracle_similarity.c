/*
 * oracle_similarity.c — CPU restatement of the reference's fingerprint-similarity and Butina paths.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or executed by the product
 * (nvmolkit_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker / the timed CPU baseline.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - Tanimoto / cosine: closed-form on integers (popcounts) followed by ONE IEEE-754 double
 *     division / sqrt, identical to RDKit's TanimotoSimilarity / CosineSimilarity on ExplicitBitVect
 *     and to the reference's SIMT kernel (src/similarity_kernels.cu:350-364).  Pinned against the
 *     reference's numpy bit-unpack restatement (nvmolkit/tests/test_clustering.py:166-180) and the
 *     hand-computed vectors in tests/golden/.
 *   - Butina: pinned by the reference's own property checker (tests/test_butina.cpp:96-153,
 *     nvmolkit/tests/test_clustering.py:23-51) and the 10x10 known answer (tests/test_butina.cpp:241-273).
 *
 * Each function cites the reference lines it follows.  Paths are into the reference repository.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_TANIMOTO 0
#define ORC_COSINE 1

static inline int popc_and(const uint32_t* a, const uint32_t* b, int W) {
  int c = 0;
  int k = 0;
  for (; k + 1 < W; k += 2) {
    uint64_t x, y;
    memcpy(&x, a + k, 8);
    memcpy(&y, b + k, 8);
    c += __builtin_popcountll(x & y);
  }
  for (; k < W; ++k) {
    c += __builtin_popcount(a[k] & b[k]);
  }
  return c;
}

static inline int popc_row(const uint32_t* a, int W) {
  int c = 0;
  for (int k = 0; k < W; ++k) {
    c += __builtin_popcount(a[k]);
  }
  return c;
}

/* src/similarity_kernels.cu:350-364 (SIMT epilogue, T_out = double). */
static inline double finish_f64(int metric, int c, int pa, int pb) {
  if (metric == ORC_TANIMOTO) {
    int u = pa + pb - c;
    if (u < 1) u = 1;
    return (double)c / (double)u;
  }
  double denom = sqrt((double)pa * (double)pb);
  return (c == 0 || denom == 0.0) ? 0.0 : (double)c / denom;
}

/* threads of the OpenMP regions of every oracle_*.c from now on (<= 0: all processors) — bench.py's one-thread baselines */
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : omp_get_num_procs());
#else
  (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static int orc_force_scalar = 0;

#if defined(__AVX512F__) && defined(__AVX512DQ__) && defined(__AVX512VPOPCNTDQ__)
#include <immintrin.h>
#define ORC_HAVE_VPOPCNT 1
/* Eight accumulators of 8 partial qword sums each -> one vector holding the eight totals, in order. */
static inline __m512i orc_reduce8(const __m512i* a) {
  const __m512i t0 = _mm512_add_epi64(_mm512_unpacklo_epi64(a[0], a[1]), _mm512_unpackhi_epi64(a[0], a[1]));
  const __m512i t1 = _mm512_add_epi64(_mm512_unpacklo_epi64(a[2], a[3]), _mm512_unpackhi_epi64(a[2], a[3]));
  const __m512i t2 = _mm512_add_epi64(_mm512_unpacklo_epi64(a[4], a[5]), _mm512_unpackhi_epi64(a[4], a[5]));
  const __m512i t3 = _mm512_add_epi64(_mm512_unpacklo_epi64(a[6], a[7]), _mm512_unpackhi_epi64(a[6], a[7]));
  const __m512i s01 = _mm512_add_epi64(_mm512_shuffle_i64x2(t0, t1, 0x88), _mm512_shuffle_i64x2(t0, t1, 0xdd));
  const __m512i s23 = _mm512_add_epi64(_mm512_shuffle_i64x2(t2, t3, 0x88), _mm512_shuffle_i64x2(t2, t3, 0xdd));
  return _mm512_add_epi64(_mm512_shuffle_i64x2(s01, s23, 0x88), _mm512_shuffle_i64x2(s01, s23, 0xdd));
}
/* Tanimoto of 8 pairs: c / max(1, pa + pb - c), IEEE double division like the scalar form. */
static inline __m512d orc_tanimoto8(const __m512i c, const int pa, const int* pb) {
  const __m512i pbv = _mm512_cvtepi32_epi64(_mm256_loadu_si256((const __m256i*)pb));
  __m512i       u   = _mm512_sub_epi64(_mm512_add_epi64(_mm512_set1_epi64(pa), pbv), c);
  u                 = _mm512_max_epi64(u, _mm512_set1_epi64(1));
  return _mm512_div_pd(_mm512_cvtepi64_pd(c), _mm512_cvtepi64_pd(u));
}
/* Rows i0, i0 + 1 (or just i0) against the 8 rows j .. j + 7: every word of B is loaded once for both rows of A.
 * NV = 512-bit words per fingerprint when known at compile time (4 at 2048 bits: the rows of A stay in registers). */
#define ORC_TILE_BODY(NVEXPR)                                                                                          \
  __m512i acc0[8], acc1[8];                                                                                            \
  const int      nv = (NVEXPR);                                                                                        \
  const __m512i* a0 = (const __m512i*)(a + i0 * W);                                                                    \
  const __m512i* a1 = (const __m512i*)(a + (i0 + (nI > 1 ? 1 : 0)) * W);                                               \
  for (int jj = 0; jj < 8; ++jj) {                                                                                     \
    const __m512i* bj = (const __m512i*)(b + (j + jj) * W);                                                            \
    __m512i        s0 = _mm512_setzero_si512(), s1 = _mm512_setzero_si512();                                           \
    for (int w = 0; w < nv; ++w) {                                                                                     \
      const __m512i bw = _mm512_loadu_si512(bj + w);                                                                   \
      s0 = _mm512_add_epi64(s0, _mm512_popcnt_epi64(_mm512_and_si512(bw, _mm512_loadu_si512(a0 + w))));                \
      s1 = _mm512_add_epi64(s1, _mm512_popcnt_epi64(_mm512_and_si512(bw, _mm512_loadu_si512(a1 + w))));                \
    }                                                                                                                  \
    acc0[jj] = s0;                                                                                                     \
    acc1[jj] = s1;                                                                                                     \
  }                                                                                                                    \
  _mm512_storeu_pd(out + i0 * ld + j, orc_tanimoto8(orc_reduce8(acc0), pa[i0], pb + j));                               \
  if (nI > 1) _mm512_storeu_pd(out + (i0 + 1) * ld + j, orc_tanimoto8(orc_reduce8(acc1), pa[i0 + 1], pb + j));
static inline void orc_tile_2x8_any(const uint32_t* a, const uint32_t* b, int W, int64_t i0, int nI, int64_t j, const int* pa,
                                    const int* pb, double* out, int64_t ld) {
  ORC_TILE_BODY(W / 16)
}
static inline void orc_tile_2x8_2048(const uint32_t* a, const uint32_t* b, int W, int64_t i0, int nI, int64_t j, const int* pa,
                                     const int* pb, double* out, int64_t ld) {
  ORC_TILE_BODY(4)
}
static inline void orc_tile_2x8(const uint32_t* a, const uint32_t* b, int W, int64_t i0, int nI, int64_t j, const int* pa,
                                const int* pb, double* out, int64_t ld) {
  if (W == 64) {
    orc_tile_2x8_2048(a, b, W, i0, nI, j, pa, pb, out, ld);
  } else {
    orc_tile_2x8_any(a, b, W, i0, nI, j, pa, pb, out, ld);
  }
}
#endif

/* out[i*ld + j] for i < nA, j < nB.  Follows launchCrossTanimotoSimilarity / launchCrossCosineSimilarity
 * (src/similarity_kernels.cu:505-582, :727-799), SIMT arithmetic.  `threads` <= 0 means all cores.
 * Where the host has AVX-512 VPOPCNTDQ (the GPU boxes' EPYC 9575F does) the Tanimoto pairs of fingerprints that are
 * whole 512-bit words run 2 x 8 at a time on vector popcounts — same integer counts, same IEEE division, checked
 * against the scalar loop in tests/test_oracle_similarity.py; the scalar loop is what every other case runs. */
void orc_cross_similarity_f64(int metric, const uint32_t* a, int64_t nA, const uint32_t* b, int64_t nB, int W,
                              double* out, int64_t ld, int threads) {
  int* pa = (int*)malloc(sizeof(int) * (size_t)(nA > 0 ? nA : 1));
  int* pb = (int*)malloc(sizeof(int) * (size_t)(nB > 0 ? nB : 1));
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
  for (int64_t i = 0; i < nA; ++i) pa[i] = popc_row(a + i * W, W);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
  for (int64_t j = 0; j < nB; ++j) pb[j] = popc_row(b + j * W, W);
  /* cache blocking only (the arithmetic per pair is unchanged): a thread owns a block of B rows that stays in its
   * cache while it sweeps blocks of A rows, so B is streamed from DRAM once per call instead of once per A row */
  const int64_t JB = 128, IB = 8;
  const int64_t nJB = (nB + JB - 1) / JB;
#ifdef ORC_HAVE_VPOPCNT
  const int vec = metric == ORC_TANIMOTO && W % 16 == 0 && W > 0 && !orc_force_scalar;
#else
  const int vec = 0;
#endif
  /* With enough rows of A for every thread, a thread owns whole blocks of output ROWS (contiguous stores, first touched by the
   * thread that keeps writing them: on a two-socket host the matrix then lies in the writer's own memory); B is re-read per
   * block of A rows from the shared cache.  Otherwise (few rows of A) the threads split the rows of B. */
  const int64_t nIB    = (nA + IB - 1) / IB;
  const int     byRows = nIB >= 2 * (int64_t)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads) if (byRows)
#endif
  for (int64_t ib = 0; ib < (byRows ? nIB : 0); ++ib) {
    const int64_t i0 = ib * IB, i1 = (i0 + IB < nA) ? i0 + IB : nA;
    for (int64_t jb = 0; jb < nJB; ++jb) {
      const int64_t j0 = jb * JB, j1 = (j0 + JB < nB) ? j0 + JB : nB;
      int64_t       jv = j0;
#ifdef ORC_HAVE_VPOPCNT
      if (vec) {
        for (; jv + 8 <= j1; jv += 8) {
          for (int64_t i = i0; i < i1; i += 2) orc_tile_2x8(a, b, W, i, (i + 1 < i1) ? 2 : 1, jv, pa, pb, out, ld);
        }
      }
#endif
      for (int64_t i = i0; i < i1; ++i) {
        const uint32_t* ai = a + i * W;
        for (int64_t j = jv; j < j1; ++j) out[i * ld + j] = finish_f64(metric, popc_and(ai, b + j * W, W), pa[i], pb[j]);
      }
    }
  }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) if (!byRows)
#endif
  for (int64_t jb = 0; jb < (byRows ? 0 : nJB); ++jb) {
    const int64_t j0 = jb * JB, j1 = (j0 + JB < nB) ? j0 + JB : nB;
    for (int64_t i0 = 0; i0 < nA; i0 += IB) {
      const int64_t i1 = (i0 + IB < nA) ? i0 + IB : nA;
      int64_t       jv = j0;
#ifdef ORC_HAVE_VPOPCNT
      if (vec) {
        for (; jv + 8 <= j1; jv += 8) {
          for (int64_t i = i0; i < i1; i += 2) orc_tile_2x8(a, b, W, i, (i + 1 < i1) ? 2 : 1, jv, pa, pb, out, ld);
        }
      }
#endif
      for (int64_t i = i0; i < i1; ++i) {
        const uint32_t* ai = a + i * W;
        for (int64_t j = jv; j < j1; ++j) {
          const int c     = popc_and(ai, b + j * W, W);
          out[i * ld + j] = finish_f64(metric, c, pa[i], pb[j]);
        }
      }
    }
  }
  free(pa);
  free(pb);
}

/* tests: 1 = keep orc_cross_similarity_f64 on the scalar loop */
void orc_set_scalar(int on) { orc_force_scalar = on; }

/* 1 if orc_cross_similarity_f64 has its vector-popcount form on this host. */
int orc_have_vpopcnt(void) {
#ifdef ORC_HAVE_VPOPCNT
  return 1;
#else
  return 0;
#endif
}

/* Integer intersection counts only (for bit-exact checks independent of the division). */
void orc_cross_intersection_i32(const uint32_t* a, int64_t nA, const uint32_t* b, int64_t nB, int W, int32_t* out) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int64_t i = 0; i < nA; ++i) {
    for (int64_t j = 0; j < nB; ++j) {
      out[i * nB + j] = popc_and(a + i * W, b + j * W, W);
    }
  }
}

/* float32 neighbour predicate of the fused path: nvmolkit/_fusedButina.py:160-173
 * (denom > 0, similarity = float(dots) / float(denom), neighbour <=> similarity >= threshold). */
static inline int is_neighbor_f32(int metric, int c, int pa, int pb, float thr) {
  float denom;
  if (metric == ORC_TANIMOTO) {
    const int u = pa + pb - c;
    if (u <= 0) return 0;
    denom = (float)u;
  } else {
    denom = sqrtf((float)pa * (float)pb);
    if (!(denom > 0.0f)) return 0;
  }
  const float sim = (float)c / denom;
  return sim >= thr;
}

/* counts[i] += sign * #{j : neighbour(x_i, y_j)}  — update_neighbor_counts, nvmolkit/_fusedButina.py:249-289. */
void orc_neighbor_counts(int metric, const uint32_t* x, int64_t nX, const uint32_t* y, int64_t nY, int W, float thr,
                         int sign, int32_t* counts) {
  int* py = (int*)malloc(sizeof(int) * (size_t)(nY > 0 ? nY : 1));
  for (int64_t j = 0; j < nY; ++j) py[j] = popc_row(y + j * W, W);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int64_t i = 0; i < nX; ++i) {
    const uint32_t* xi = x + i * W;
    const int       px = popc_row(xi, W);
    int             n  = 0;
    for (int64_t j = 0; j < nY; ++j) {
      n += is_neighbor_f32(metric, popc_and(xi, y + j * W, W), px, py[j], thr);
    }
    counts[i] += sign * n;
  }
  free(py);
}

/*
 * Matrix-free Taylor-Butina — fused_butina, nvmolkit/clustering.py:99-189 with
 * nvmolkit/_fusedButina.py:182-246 for the per-round extraction.
 *
 * Stated directly on the definition (O(N^2) adjacency bits, so for small N only):
 *   repeat: degree[i] = #free neighbours of free row i (self included when its own denom > 0);
 *           stop when max degree == 0; centroid = LAST row with the max degree (clustering.py:159);
 *           cluster = centroid + its free neighbours; additionally every free non-member whose
 *           degree is 1 is harvested as a singleton (_fusedButina.py:240-244).
 * Output convention of this build (the reference's member order comes from atomics and is
 * unspecified): greedy clusters in the order found, centroid first then members ascending;
 * singleton tail ascending by row.  Rows with degree 0 (all-zero fingerprints) are appended to the
 * singleton tail (the reference leaves them unwritten — clustering.py:171-175 reads zeros there).
 * Returns the number of clusters; offsets has n_clusters+1 entries.
 */
static int64_t butina_rounds_on_adjacency(const uint8_t* adj, int64_t N, int32_t* cluster_indices, int64_t* offsets,
                                          int32_t* centroids);

int64_t orc_butina_fused(int metric, const uint32_t* x, int64_t N, int W, double cutoff, int32_t* cluster_indices,
                         int64_t* offsets, int32_t* centroids) {
  const float thr = (float)(1.0 - cutoff); /* clustering.py:149, passed to the kernel as f32 */
  uint8_t*    adj = (uint8_t*)calloc((size_t)(N * N > 0 ? N * N : 1), 1);
  int*        pc  = (int*)malloc(sizeof(int) * (size_t)(N > 0 ? N : 1));
  for (int64_t i = 0; i < N; ++i) pc[i] = popc_row(x + i * W, W);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int64_t i = 0; i < N; ++i) {
    for (int64_t j = 0; j < N; ++j) {
      adj[i * N + j] = (uint8_t)is_neighbor_f32(metric, popc_and(x + i * W, x + j * W, W), pc[i], pc[j], thr);
    }
  }
  const int64_t nc = butina_rounds_on_adjacency(adj, N, cluster_indices, offsets, centroids);
  free(adj);
  free(pc);
  return nc;
}

/* The same round loop on a neighbour GRAPH: `pairs` lists every unordered neighbour pair (i != j) once, `counts[i]` is
 * row i's degree INCLUDING itself when it is its own neighbour (an all-zero fingerprint is not) — the inputs of the
 * product's nvmk_butina_from_pairs (row-sharded fused Butina). */
int64_t orc_butina_from_pairs(int64_t N, const int32_t* counts, const int32_t* pairs, int64_t n_pairs, int32_t* cluster_indices,
                              int64_t* offsets, int32_t* centroids) {
  uint8_t* adj = (uint8_t*)calloc((size_t)(N * N > 0 ? N * N : 1), 1);
  int32_t* inc = (int32_t*)calloc((size_t)(N > 0 ? N : 1), sizeof(int32_t));
  for (int64_t e = 0; e < n_pairs; ++e) {
    const int64_t i = pairs[2 * e], j = pairs[2 * e + 1];
    adj[i * N + j] = adj[j * N + i] = 1;
    ++inc[i];
    ++inc[j];
  }
  for (int64_t i = 0; i < N; ++i) adj[i * N + i] = (uint8_t)(counts[i] - inc[i] > 0);
  const int64_t nc = butina_rounds_on_adjacency(adj, N, cluster_indices, offsets, centroids);
  free(adj);
  free(inc);
  return nc;
}

static int64_t butina_rounds_on_adjacency(const uint8_t* adj, int64_t N, int32_t* cluster_indices, int64_t* offsets,
                                          int32_t* centroids) {
  uint8_t* is_free   = (uint8_t*)malloc((size_t)(N > 0 ? N : 1));
  int32_t* degree    = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  int32_t* singles   = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  int64_t  nSingles  = 0;
  int64_t  nClusters = 0;
  int64_t  pos       = 0;
  memset(is_free, 1, (size_t)N);
  offsets[0] = 0;
  for (;;) {
    int32_t best = 0;
    int64_t arg  = -1;
    for (int64_t i = 0; i < N; ++i) {
      degree[i] = 0;
      if (!is_free[i]) continue;
      int32_t d = 0;
      for (int64_t j = 0; j < N; ++j) d += (is_free[j] && adj[i * N + j]);
      degree[i] = d;
      if (d >= best && d > 0) {
        best = d;
        arg  = i;
      }
    }
    if (arg < 0) break;
    centroids[nClusters]   = (int32_t)arg;
    cluster_indices[pos++] = (int32_t)arg;
    for (int64_t j = 0; j < N; ++j) {
      if (j != arg && is_free[j] && adj[arg * N + j]) cluster_indices[pos++] = (int32_t)j;
    }
    /* singleton harvest uses the degrees of THIS round (before removal) */
    for (int64_t j = 0; j < N; ++j) {
      if (is_free[j] && j != arg && !adj[arg * N + j] && degree[j] == 1) {
        singles[nSingles++] = (int32_t)j;
        is_free[j]          = 0;
      }
    }
    for (int64_t j = 0; j < N; ++j) {
      if (adj[arg * N + j]) is_free[j] = 0;
    }
    is_free[arg]         = 0;
    offsets[++nClusters] = pos;
  }
  for (int64_t j = 0; j < N; ++j) {
    if (is_free[j]) singles[nSingles++] = (int32_t)j; /* degree-0 rows */
  }
  /* ascending singleton tail */
  for (int64_t a = 1; a < nSingles; ++a) {
    int32_t v = singles[a];
    int64_t b = a - 1;
    while (b >= 0 && singles[b] > v) {
      singles[b + 1] = singles[b];
      --b;
    }
    singles[b + 1] = v;
  }
  for (int64_t s = 0; s < nSingles; ++s) {
    centroids[nClusters]   = singles[s];
    cluster_indices[pos++] = singles[s];
    offsets[++nClusters]   = pos;
  }
  free(is_free);
  free(degree);
  free(singles);
  return nClusters;
}

/*
 * Taylor-Butina on a dense matrix — butinaGpu, src/butina.cu:913-1071.
 *   hit[i][j] = dist[i][j] <= cutoff                      (:1043-1051)
 *   loop while the largest unassigned neighbourhood has >= 2 members (kMinLoopSizeForAssignment, :34):
 *     centroid = LAST index with the largest count (lastArgMaxKernel :464-481), assign it and its
 *     unassigned neighbours the next cluster id (:241-269);
 *   remaining points become singletons in ascending index order (:281-307 — the reference's order is
 *   an atomic race; ascending is this build's convention);
 *   renumber: id 0 = largest cluster, stable by original id (:369-448).
 * `hit` may be NULL (then dist/cutoff are used) or dist may be NULL (then hit is used).
 */
int64_t orc_butina_dense(const double* dist, const uint8_t* hit_in, int64_t N, double cutoff, int32_t* clusters,
                         int32_t* centroids_out) {
  uint8_t* hit = (uint8_t*)malloc((size_t)(N * N > 0 ? N * N : 1));
  for (int64_t k = 0; k < N * N; ++k) hit[k] = dist ? (uint8_t)(dist[k] <= cutoff) : (uint8_t)(hit_in[k] != 0);
  int32_t* cent = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  for (int64_t i = 0; i < N; ++i) clusters[i] = -1;
  int64_t next = 0;
  for (;;) {
    int32_t best = -1;
    int64_t arg  = -1;
    for (int64_t i = 0; i < N; ++i) {
      int32_t d = 0;
      if (clusters[i] < 0) {
        for (int64_t j = 0; j < N; ++j) d += (hit[i * N + j] && clusters[j] < 0);
      }
      if (d >= best) {
        best = d;
        arg  = i;
      }
    }
    if (best < 2) break;
    for (int64_t j = 0; j < N; ++j) {
      if (hit[arg * N + j] && clusters[j] < 0) clusters[j] = (int32_t)next;
    }
    clusters[arg] = (int32_t)next;
    cent[next++]  = (int32_t)arg;
  }
  for (int64_t i = 0; i < N; ++i) {
    if (clusters[i] < 0) {
      clusters[i]  = (int32_t)next;
      cent[next++] = (int32_t)i;
    }
  }
  /* renumber by descending size, stable by original id */
  int64_t  nC    = next;
  int64_t* sizes = (int64_t*)calloc((size_t)(nC > 0 ? nC : 1), sizeof(int64_t));
  int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nC > 0 ? nC : 1));
  int32_t* remap = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nC > 0 ? nC : 1));
  for (int64_t i = 0; i < N; ++i) sizes[clusters[i]]++;
  for (int64_t c = 0; c < nC; ++c) order[c] = c;
  /* stable insertion sort is O(nC^2); use a counting pass by size instead */
  {
    int64_t maxSize = 0;
    for (int64_t c = 0; c < nC; ++c)
      if (sizes[c] > maxSize) maxSize = sizes[c];
    int64_t* start = (int64_t*)calloc((size_t)(maxSize + 2), sizeof(int64_t));
    for (int64_t c = 0; c < nC; ++c) start[maxSize - sizes[c] + 1]++;
    for (int64_t s = 1; s <= maxSize + 1; ++s) start[s] += start[s - 1];
    for (int64_t c = 0; c < nC; ++c) order[start[maxSize - sizes[c]]++] = c;
    free(start);
  }
  for (int64_t newId = 0; newId < nC; ++newId) remap[order[newId]] = (int32_t)newId;
  for (int64_t i = 0; i < N; ++i) clusters[i] = remap[clusters[i]];
  if (centroids_out) {
    for (int64_t newId = 0; newId < nC; ++newId) centroids_out[newId] = cent[order[newId]];
  }
  free(hit);
  free(cent);
  free(sizes);
  free(order);
  free(remap);
  return nC;
}

/*
 * Exhaustive check of the division shortcut used by the HIP epilogue (not reference arithmetic: it
 * validates an implementation shortcut against IEEE division).  The device computes
 *   r0 = v_rcp_f32((float)u)          (hardware reciprocal, accurate to 1 ulp of f32)
 *   r  = fma(fma(-u, r0, 1), r0, r0)  (one Newton step in f64)
 *   q0 = RN(c * r);  e = fma(-q0, u, c);  q = fma(e, r, q0)
 * A 1-ulp reciprocal is one of the three floats around RN_f32(1 / u), so all three seeds are tried.
 * Returns the number of (seed, c, u) triples, 0 <= c <= u, 1 <= u <= umax, for which q != (double)c / (double)u.
 */
int64_t orc_check_newton_division(int umax) {
  int64_t bad = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : bad)
#endif
  for (int u = 1; u <= umax; ++u) {
    const double ud    = (double)u;
    const float  rn    = 1.0f / (float)u;
    const float  seeds[3] = {nextafterf(rn, 0.0f), rn, nextafterf(rn, 2.0f)};
    for (int k = 0; k < 3; ++k) {
      const double r0 = (double)seeds[k];
      const double r  = fma(fma(-ud, r0, 1.0), r0, r0);
      for (int c = 0; c <= u; ++c) {
        const double cd = (double)c;
        const double q0 = cd * r;
        const double e  = fma(-q0, ud, cd);
        const double q  = fma(e, r, q0);
        if (q != cd / ud) ++bad;
      }
    }
  }
  return bad;
}

/* Table-free form of the fused-Butina neighbour predicate used by the matrix-core count kernels
 * (nvmolkit_amd/csrc/similarity_mfma.hip, arith_threshold / ARITH epilogues):
 *   (float)c / (float)(s - c) >= thr   <=>   c (1 + m) - pb m  >  pa m - adj,      s = pa + pb,
 * m = midpoint of thr and its float predecessor, adj = half a grid unit when the tie at m rounds up to thr (even
 * significand) and 0 otherwise, all in exact double arithmetic.
 * Returns the number of (s, c) pairs, 1 <= s <= smax, 0 <= c < s, on which the two predicates differ (must be 0).
 */
int64_t orc_check_threshold_arith(float thr, int smax) {
  const float  pred = nextafterf(thr, -INFINITY);
  const double m    = 0.5 * ((double)thr + (double)pred);
  const double grid = 0.5 * ((double)thr - (double)pred);
  uint32_t     bits;
  memcpy(&bits, &thr, sizeof(bits));
  const double adj = (bits & 1u) ? 0.0 : 0.5 * grid;
  const double k1 = 1.0 + m, k2 = m;
  int64_t      bad = 0;
  for (int s = 1; s <= smax; ++s) {
    for (int c = 0; c < s; ++c) {
      const int ref = (float)c / (float)(s - c) >= thr;
      /* the kernel keeps the row and column parts of s apart: any split must decide the same */
      const int    pa  = s / 3, pb = s - pa;
      const double paK = fma((double)pa, k2, -adj);
      const double d   = fma((double)c, k1, -((double)pb * k2));
      if (ref != (d > paK)) ++bad;
    }
  }
  return bad;
}
