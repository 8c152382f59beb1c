/*
 * nvmolkit_amd.h — C ABI of the MI355X (gfx950) hot-path library `libnvmolkit_amd.so`.
 *
 * This is the drop-in boundary for the batched data-parallel hot path of nvMolKit
 * (fingerprint similarity / Butina clustering / Morgan fingerprints / force fields /
 * BFGS / ETKDG).  The reference has no C ABI: its Boost.Python modules call C++
 * directly (see INTEGRATION.md for the binding a maintainer would add).  Every entry
 * point below names the reference interface it replaces (paths are into the reference
 * repository, `file:line`).
 *
 * Conventions (all entry points):
 *   - return 0 on success, a negative NVMK_ERR_* code on failure; the message of the
 *     most recent failure on the calling thread is returned by nvmk_last_error();
 *   - no exception crosses the boundary, no torch / C++ types appear in a signature;
 *   - pointers named d_* are DEVICE pointers valid on the current HIP device, h_* are
 *     HOST pointers; the library never allocates a caller-visible output;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); work is
 *     enqueued asynchronously on it unless the function is documented as blocking;
 *   - callable concurrently from different host threads on different streams.
 */
#ifndef NVMOLKIT_AMD_H
#define NVMOLKIT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVMK_OK 0
#define NVMK_ERR_INVALID_ARGUMENT (-1) /* maps to Python ValueError (reference: std::invalid_argument) */
#define NVMK_ERR_HIP (-2)              /* HIP runtime failure (reference: CudaBadReturnCode, src/utils/cuda_error_check.h:28-46) */
#define NVMK_ERR_OUT_OF_MEMORY (-3)    /* reference: std::runtime_error("Not enough memory…"), src/similarity.cpp:137-139 */
#define NVMK_ERR_UNSUPPORTED (-4)
#define NVMK_ERR_INTERNAL (-5)
#define NVMK_TRUNCATED 1 /* success with a notice: the output is valid but incomplete (only where an entry point says so) */

/* Similarity metric selector (reference: enum SimilarityType, src/similarity_kernels.cu). */
#define NVMK_METRIC_TANIMOTO 0
#define NVMK_METRIC_COSINE 1

/* ---- library / device info ------------------------------------------------------------------ */

/* Message of the last failure on this thread ("" if none).  Never NULL. */
const char* nvmk_last_error(void);
/* ABI version of this header: (major << 16) | minor. */
int nvmk_abi_version(void);
/* Number of visible HIP devices (reference: src/utils/device.h countCudaDevices). */
int nvmk_device_count(int* count);
/* Free / total bytes on the current device (reference: getDeviceFreeMemory, src/utils/device.h). */
int nvmk_device_memory(size_t* free_bytes, size_t* total_bytes);

/* ---- multi-GPU contract: device set, peer copies, the one collective of the path -------------------------------------
 * One process per GPU is the scaling model (BASELINE.json: molecule batches shard with no collective; only the N x M
 * cross-similarity assembles the sharded reference fingerprints with an all-gather over xGMI), one process driving several
 * GPUs the reference's (BatchHardwareOptions::gpuIds, src/hardware_options.h:26-35).  Both are served:
 *
 * nvmk_set_devices : the GPUs this process works on (n = 0: all visible) with peer access enabled between every pair, both
 *                    ways, idempotent (reference: enablePeerAccess, src/utils/p2p.cpp:30-58).  nvmk_get_devices reads the set.
 * nvmk_copy_peer_async : d_src on src_device (ordered after src_stream's work so far) -> d_dst on dst_device, on dst_stream
 *                    (reference: copyDeviceToDeviceAsync, src/utils/p2p.cpp:60-86; caller device_coord_collector.cpp:86-109,
 *                    which stitches per-GPU conformer blocks on the target GPU).
 * nvmk_allgather_rows : every rank contributes rows_per_rank rows of words_per_row uint32 (packed fingerprints) and receives
 *                    all ranks' rows in rank order: ncclAllGather over RCCL on `stream` (asynchronous).  `comm` is an
 *                    ncclComm_t — the caller's own, or one made with the three helpers below (thin wrappers of
 *                    ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy; the 128-byte id travels from rank 0 to the
 *                    others by whatever means the host has: MPI, a file, a socket).  RCCL is resolved at run time from the
 *                    process (a communicator belongs to the RCCL copy that made it) or from librccl.so.1; the library
 *                    does not link against it.  The reference has no counterpart: it drives every GPU from one process. */
int nvmk_set_devices(const int32_t* device_ids, int n);
int nvmk_get_devices(int32_t* device_ids, int capacity, int* n);
int nvmk_copy_peer_async(void* d_dst, int dst_device, void* dst_stream, const void* d_src, int src_device, void* src_stream,
                         size_t bytes);
int nvmk_comm_unique_id(char id[128]);
int nvmk_comm_init_rank(void** comm, int n_ranks, const char id[128], int rank);
int nvmk_comm_destroy(void* comm);
int nvmk_allgather_rows(void* comm, const uint32_t* d_send, int64_t rows_per_rank, int words_per_row, uint32_t* d_recv,
                        void* stream);
/* Tuning / test switches (the NVMK_* names of DESIGN.md section 5: NVMK_SIM_PATH, NVMK_BFGS_LDS, ...).  The environment is
 * read ONCE per process, when the first switch is looked up; afterwards a switch changes only through nvmk_set_option
 * (value NULL or "" = unset), which is safe against concurrent callers of the other entry points: each call takes one
 * consistent snapshot.  The reference has no counterpart (its tuning lives in compile-time constants and
 * BatchHardwareOptions, src/utils/host_vector.h / nvmolkit/types.py:44-110).  Unknown names are an invalid argument. */
int nvmk_set_option(const char* name, const char* value);
int nvmk_get_option(const char* name, char* value, size_t capacity);

/* ---- S1/S2: dense N x M cross-similarity ------------------------------------------------------
 * Replaces launchCrossTanimotoSimilarity / launchCrossCosineSimilarity
 * (src/similarity_kernels.cu:505-582, :727-799) and their callers
 * crossTanimotoSimilarityGpuResult / crossCosineSimilarityGpuResult (src/similarity.cpp:38-58, :260-280).
 *
 *   d_a   : nA rows of `fp_bits/32` uint32 words, row-major (bit j of a fingerprint = bit j%32 of word j/32)
 *   d_b   : nB rows, same width
 *   d_out : nA x nB doubles, row stride `ld_out` elements (pass nB for a dense matrix)
 *
 *   tanimoto: out[i][j] = c / max(1, pa + pb - c),   c = popcount(a_i & b_j)   (double division)
 *   cosine  : out[i][j] = (c == 0 || pa*pb == 0) ? 0 : c / sqrt((double)pa * (double)pb)
 * (the reference's SIMT formulas, src/similarity_kernels.cu:350-364).
 * fp_bits must be a positive multiple of 32.  nA == 0 or nB == 0 is a no-op.
 */
int nvmk_cross_tanimoto_f64(const uint32_t* d_a, int64_t nA, const uint32_t* d_b, int64_t nB, int fp_bits,
                            double* d_out, int64_t ld_out, void* stream);
int nvmk_cross_cosine_f64(const uint32_t* d_a, int64_t nA, const uint32_t* d_b, int64_t nB, int fp_bits,
                          double* d_out, int64_t ld_out, void* stream);

/* ---- S1 (matrix-core formulation): prepared FP4 fingerprint sets ------------------------------------
 * The reference computes popcount(a & b) on NVIDIA's 1-bit tensor-core MMA (src/similarity_kernels.cu:96-240,
 * src/utils/macros_ptx.cuh:137-211).  The MI355X counterpart expands every fingerprint bit once into an FP4
 * (e2m1) nibble and runs the N x M x K work on v_mfma_scale_f32_32x32x64_f8f6f4 (exact: products are 0/1, f32
 * accumulation is exact below 2^24).  nvmk_cross_tanimoto_f64 / nvmk_cross_cosine_f64 do this internally;
 * these entry points let a caller that reuses a fingerprint set (row-chunked 1M x 1M, Butina rounds) expand it once.
 *   nvmk_fp4_workspace_bytes : device bytes needed for a prepared set of n fingerprints (0 on bad arguments)
 *   nvmk_fp4_prepare         : expand d_in (n x fp_bits/32 words) into d_workspace (also stores row popcounts)
 *   nvmk_cross_similarity_prepared_f64 : rows [a_row0, a_row0 + a_rows) of prepared set A (a_row0 % 128 == 0)
 *                              against all nB rows of prepared set B; output as nvmk_cross_tanimoto_f64.
 */
size_t nvmk_fp4_workspace_bytes(int64_t n, int fp_bits);
int nvmk_fp4_prepare(const uint32_t* d_in, int64_t n, int fp_bits, void* d_workspace, void* stream);
int nvmk_cross_similarity_prepared_f64(int metric, const void* d_ws_a, int64_t nA_total, int64_t a_row0, int64_t a_rows,
                                       const void* d_ws_b, int64_t nB, int fp_bits, double* d_out, int64_t ld_out,
                                       void* stream);

/* ---- S4: memory-constrained cross-similarity returned on the host --------------------------------
 * Replaces crossTanimotoSimilarityCPUResult / crossCosineSimilarityCPUResult -> crossSimilarityImpl
 * (src/similarity.cpp:105-254, :282-297; CrossSimilarityOptions src/similarity.h:29-32).
 * Blocking.  `h_out` is nA*nB doubles on the host (any pageable or pinned memory).
 * max_device_bytes < 0 means "use the free memory of the current device".  When the matrix does
 * not fit, rows of `a` are processed in chunks of max(32, floor(((max/2)*0.9/8)/(32*nB))*32) rows on two
 * streams with double-buffered pinned staging; if 32 rows do not fit -> NVMK_ERR_OUT_OF_MEMORY.
 */
int nvmk_cross_similarity_host_f64(int metric, const uint32_t* d_a, int64_t nA, const uint32_t* d_b, int64_t nB,
                                   int fp_bits, double* h_out, int64_t max_device_bytes);

/* ---- B2 building blocks: matrix-free neighbour counting -----------------------------------------
 * Replaces update_neighbor_counts / _update_neighbor_count_kernel (nvmolkit/_fusedButina.py:99-179, :249-289).
 * For every row i of x: counts[i] += sign * #{ j : float32(sim(x_i, y_j)) >= thr_f32 and denom > 0 }.
 * sim is evaluated in float32 exactly as the reference does (float(c) / float(denom)).
 * sign must be +1 or -1.  d_x_rows / d_y_rows are optional row-index lists (NULL = rows 0..n-1):
 * the logical row r of x is the physical row d_x_rows[r] of d_x, which lets a caller keep the
 * fingerprint matrix in place instead of compacting it every round.  counts is indexed by the
 * PHYSICAL row (counts[d_x_rows[r]]), i.e. it stays aligned with d_x across rounds.
 * d_y == d_x with nY == nX and no row lists (a set against itself, the first pass of fused Butina) is recognised: only the
 * pairs on or above the diagonal are evaluated and each credits both rows — the same counts for half the work.
 */
int nvmk_neighbor_counts(int metric, const uint32_t* d_x, const int32_t* d_x_rows, int64_t nX, const uint32_t* d_y,
                         const int32_t* d_y_rows, int64_t nY, int fp_bits, float threshold, int sign,
                         int32_t* d_counts, void* stream);

/* ---- B2: fused (matrix-free) Taylor-Butina -------------------------------------------------------
 * Replaces fused_butina (nvmolkit/clustering.py:99-189).  Blocking.
 *   d_x              : N x fp_bits/32 words
 *   cutoff           : distance cutoff in [0,1]; neighbour <=> float32(sim) >= float32(1 - cutoff)
 *   h_cluster_indices: N int32; members grouped by cluster, cluster k = [h_offsets[k], h_offsets[k+1]),
 *                      centroid FIRST inside each group
 *   h_offsets        : N+1 int64 capacity; first *n_clusters+1 entries valid (h_offsets[0] = 0),
 *                      i.e. the reference's cumulative `cluster_sizes` list
 *   h_centroids      : N int32 capacity, first *n_clusters entries valid
 * Cluster sizes are non-increasing up to the singleton tail, ties are broken toward the highest
 * row index (clustering.py:159).
 */
int nvmk_butina_fused(int metric, const uint32_t* d_x, int64_t N, int fp_bits, double cutoff,
                      int32_t* h_cluster_indices, int64_t* h_offsets, int32_t* h_centroids, int64_t* n_clusters,
                      void* stream);

/* Fused Butina in two steps, for row-sharding the all-pairs pass over several GPUs (the reference is
 * single-GPU here: nvmolkit/clustering.py:99-189).
 *   nvmk_butina_pairs      : shard `shard` of `n_shards` evaluates its band of 128-row tile rows of the symmetric pass
 *                            (bands of equal area).  d_counts (N int32, overwritten) receives the PARTIAL neighbour counts of
 *                            all rows from those tiles; d_pairs (2 * pair_capacity int32) the neighbour pairs (i, j) found,
 *                            each once, original row numbers; *h_n_pairs their number.  NVMK_ERR_OUT_OF_MEMORY when they do
 *                            not fit (*h_n_pairs still says how many there were).
 *   nvmk_butina_from_pairs : the clustering from the full degrees (sum of the shards' counts) and the concatenated pairs
 *                            — the same device-side round loop nvmk_butina_fused runs; outputs as nvmk_butina_fused.
 * One shard (shard 0 of 1) followed by from_pairs gives exactly nvmk_butina_fused's clusters; every rank that runs
 * from_pairs on the assembled graph gets the same answer, so no per-round collective is needed. */
int nvmk_butina_pairs(int metric, const uint32_t* d_x, int64_t N, int fp_bits, double cutoff, int shard, int n_shards,
                      int32_t* d_counts, int32_t* d_pairs, uint64_t pair_capacity, uint64_t* h_n_pairs, void* stream);
int nvmk_butina_from_pairs(int64_t N, const int32_t* d_counts, const int32_t* d_pairs, uint64_t n_pairs, int32_t* h_cluster_indices,
                           int64_t* h_offsets, int32_t* h_centroids, int64_t* n_clusters, void* stream);

/* ---- B1: Taylor-Butina on a dense matrix ----------------------------------------------------------
 * Replaces butinaGpu(span<const double>, ...) and butinaGpu(span<const uint8_t>, ...)
 * (src/butina.cu:1017-1071).  Blocking (the reference also synchronises the stream before returning,
 * src/butina.cu:1006-1014).
 *   d_dist      : N x N doubles (neighbour <=> dist <= cutoff, src/butina.cu:1043-1051), or NULL
 *   d_hit       : N x N uint8 adjacency (used when d_dist is NULL)
 *   d_clusters  : N int32 out; cluster ids, id 0 = largest cluster, ties by ascending original id
 *   d_centroids : N int32 capacity out or NULL; centroid row of each cluster id
 *   neighborlist_max_size : must be one of 8 / 16 / 24 / 32 / 64 / 128 like the reference (nvmolkit/clustering.py:79-82) and
 *                 is otherwise IGNORED: it sizes the neighbour lists of the reference's small-cluster phase
 *                 (src/butina.cu:975-1004); this implementation has one phase (byte hit matrix + its transpose, contiguous
 *                 row / column reads every round), so there is nothing for it to tune.  Results do not depend on it in the
 *                 reference either.
 * Singleton ids and the renumbering by size (src/butina.cu:281-307, :369-448) run on the device.
 */
int nvmk_butina_dense(const double* d_dist, const uint8_t* d_hit, int64_t N, double cutoff, int neighborlist_max_size,
                      int32_t* d_clusters, int32_t* d_centroids, int64_t* h_n_clusters, void* stream);

/* ---- M2: Morgan fingerprints from flattened invariants ----------------------------------------------
 * Replaces launchMorganFingerprintKernelBatch / morganFingerprintKernelBatch<maxAtoms, fpSize>
 * (src/morgan_fingerprint_kernels.cu:151-485; buffers MorganGPUBuffersBatch, morgan_fingerprint_kernels.h:30-42).
 * Inputs are the arrays produced by MorganInvariantsGenerator::ComputeInvariantsInto
 * (src/morgan_fingerprint_common.cpp:43-124), one slot of `max_atoms` entries per molecule:
 *   d_atom_inv   [n_mols*max_atoms] u32   atom invariants
 *   d_bond_inv   [n_mols*max_atoms] u32   bond invariants (bond type), indexed by bond id
 *   d_bond_idx   [n_mols*max_atoms*8] i16 bond ids incident to each atom, -1 padded
 *   d_bond_other [n_mols*max_atoms*8] i16 neighbour atom id across that bond, -1 padded
 *   d_n_atoms    [n_mols] i16
 *   d_out_idx    [n_mols] i32            output row of each molecule (NULL = identity)
 *   d_out        rows of fp_bits/32 u32  (rows are OVERWRITTEN for the listed molecules)
 * max_atoms in {32, 64, 128, 256, 512, 1024} (the reference computes molecules of >= 128 atoms on the CPU,
 * src/morgan_fingerprint_gpu.cpp:181-188, :296-304; here they stay on the GPU: up to 256 with all state in LDS, the 512 and
 * 1024 buckets with the neighbourhood bitsets in a stream-ordered global scratch);
 * fp_bits in {128, 256, 512, 1024, 2048, 4096}; radius in [0, 8].  Atoms and bonds of a molecule must both be
 * < max_atoms (the reference's bucketing rule, src/morgan_fingerprint_common.cpp:71-73).
 */
int nvmk_morgan_from_invariants(const uint32_t* d_atom_inv, const uint32_t* d_bond_inv, const int16_t* d_bond_idx,
                                const int16_t* d_bond_other, const int16_t* d_n_atoms, const int32_t* d_out_idx,
                                int64_t n_mols, int max_atoms, int radius, int fp_bits, uint32_t* d_out, void* stream);

/* ---- F1-F4 / N1-N2: batched force fields and the fused BFGS minimiser ---------------------------------
 * Replaces the BatchedForcefield interface (src/forcefields/batched_forcefield.h:74-149: computeEnergy /
 * computeGradients with an active-system mask), the block-per-molecule kernels (src/forcefields/mmff_kernels.cu:
 * 1064-1157, dist_geom_kernels_device.cuh:832-1355) and the fused per-molecule BFGS
 * (src/minimizer/bfgs_minimize_permol_kernels.cu:426-745; BfgsBatchMinimizer::minimize, bfgs_minimize.cu:978-1084).
 *
 * A batch is `n_systems` independent systems (conformers); system s owns atoms [atom_starts[s], atom_starts[s+1])
 * of the position array, `dim` doubles per atom (dim = 3 for ETK / MMFF / UFF, 4 for DG / QUARTIC; the reference keeps ETK 4-D, but its
 * terms never touch the 4th coordinate).  Term tables are the
 * flattened arrays the reference builds in rdkit_extensions/ (SoA + CSR): group g holds the terms of one type for
 * all systems, terms of system s are [starts[s], starts[s+1]); `idx` has n_idx LOCAL atom indices per term and
 * `par` n_par doubles per term, both interleaved per term:
 *   NVMK_FF_DG   (src/forcefields/dist_geom.h:31-56)    w0 = chiral weight, w1 = fourth-dimension weight
 *     g0 distance violation  idx(i, j)        par(lb2, ub2, weight)
 *     g1 chiral volume       idx(1, 2, 3, 4)  par(volLower, volUpper)
 *     g2 fourth dimension    idx(i)           -
 *   NVMK_FF_ETK  (dist_geom.h:73-130)
 *     g0 experimental torsion idx(1..4) par(fc[6], sign[6])      g1 inversion idx(1..4) par(C0, C1, C2, k)
 *     g2 1-2 distance idx(i, j) par(minLen, maxLen, k, pinned)   g3 1-3 distance (same; pinned = isImproperConstrained)
 *     g4 1-3 angle idx(1, 2, 3) par(minAngle, maxAngle) [deg]    g5 long-range distance (as g2)
 *   NVMK_FF_MMFF (src/forcefields/mmff.h:37-145)
 *     g0 bond idx(i, j) par(r0, kb)                 g1 angle idx(1,2,3) par(theta0, ka, isLinear)
 *     g2 stretch-bend idx(1,2,3) par(theta0, r0ij, r0kj, kbaIJK, kbaKJI)   g3 out-of-plane idx(1..4) par(koop)
 *     g4 torsion idx(1..4) par(V1, V2, V3)          g5 vdW idx(i, j) par(R*, eps)
 *     g6 electrostatic idx(i, j) par(qi*qj/D, dielModel, is1_4)
 *     g11 (optional) merged non-bonded pairs idx(i, j) par(R*, eps, qi*qj/D, dielModel, is1_4): one row per vdW pair with
 *         the electrostatic parameters of the same pair (0 charge term where there is none).  When present and the mask
 *         enables both g5 and g6, it is evaluated INSTEAD of them: one distance, one set of force accumulations per pair
 *         (the Python layer builds it on the device whenever every g6 pair is also a g5 pair, which is how RDKit and the
 *         reference's builder emit them: rdkit_extensions/mmff_flattened_builder.cpp addVdW / addEle)
 *   NVMK_FF_UFF  (src/forcefields/uff.h:27-67; term math src/forcefields/uff_kernels_device.cuh:37-580)
 *     g0 bond idx(i, j) par(restLen, k)             g1 angle idx(1,2,3) par(theta0, k, order, C0, C1, C2)
 *     g2 torsion idx(1..4) par(k, order, cosTerm)   g3 inversion idx(1..4) par(k, C0, C1, C2)
 *     g4 vdW idx(i, j) par(x_ij, wellDepth, threshold)
 *   Constraint groups (optional, MMFF: g7..g10, UFF: g5..g8; NULL starts = none; reference terms
 *   src/forcefields/mmff_kernels_device.cuh:663-1036, SoA mmff.h:101-145):
 *     distance idx(i, j) par(minLen, maxLen, k)         E = k/2 d^2 outside [minLen, maxLen]
 *     position idx(i) par(x, y, z, maxDispl, k)          E = k/2 max(|p - ref| - maxDispl, 0)^2
 *     angle idx(1, 2, 3) par(minDeg, maxDeg, k)          E = k dTheta^2 [deg^2]
 *     torsion idx(1..4) par(minDeg, maxDeg, k)           E = k dPhi^2 [deg^2], signed dihedral, periodic window
 *   NVMK_FF_QUARTIC: the synthetic field of the reference's BFGS tests (tests/test_bfgs_minimizer.cu:823-860),
 *     E = sum (x_p - p)^4 over global coordinate index p; w0 != 0 includes every atom's 4th coordinate.
 * All pointers inside the struct are DEVICE pointers; the struct itself is passed by host pointer.
 */
#define NVMK_FF_DG 0
#define NVMK_FF_ETK 1
#define NVMK_FF_MMFF 2
#define NVMK_FF_QUARTIC 3
#define NVMK_FF_UFF 4

typedef struct nvmk_ff_group {
  const int32_t* starts; /* [n_systems + 1] */
  const int32_t* idx;
  const double*  par;
} nvmk_ff_group;

typedef struct nvmk_ff_batch {
  int32_t        kind;
  int32_t        n_systems;
  const int32_t* atom_starts; /* device, [n_systems + 1] */
  nvmk_ff_group  groups[12];
  /* Optional (NULL / 0 = unused).  system_mol: the term tables are per MOLECULE and system s reads row
   * system_mol[s] of every `starts` array — conformers of one molecule share one copy of the tables (the
   * reference replicates them per conformer, src/forcefields/mmff.h:327-344).  group_mask: bit g enables group g
   * (0 = all; used for the ETK planarity energy = impropers only).  etk_ref12/13: per-system reference distances
   * of the ETK 1-2 / 1-3 restraints, laid out like the system's terms; bounds become ref +- (max - min) / 2
   * (updateReferencePositionsKernel, src/etkdg_stage_etk_minimization.cu:32-64). */
  const int32_t* system_mol;
  uint32_t       group_mask;
  const int32_t* etk_ref12_starts;
  const double*  etk_ref12;
  const int32_t* etk_ref13_starts;
  const double*  etk_ref13;
} nvmk_ff_batch;

/* energies[s] / gradient (same layout as d_pos) of every system with d_active[s] != 0 (NULL = all). */
int nvmk_ff_energy(const nvmk_ff_batch* batch, double w0, double w1, const double* d_pos, const uint8_t* d_active,
                   double* d_energies, void* stream);
int nvmk_ff_gradient(const nvmk_ff_batch* batch, double w0, double w1, const double* d_pos, const uint8_t* d_active,
                     double* d_grad, void* stream);
/* Minimises every active system in place.  h_atom_starts is a HOST copy of atom_starts (sizes the per-system
 * inverse Hessians).  d_statuses[s] = 0 converged / 1 not (reference: statuses_, 0 == converged), d_iters optional.
 * Blocking.  Constants as the reference (FUNCTOL 1e-4, MOVETOL 1e-7, TOLX 1.2e-7, EPS 3e-8, <= 1000 line-search steps).
 * Any system size (the reference: shared-memory and global-memory instantiations, bfgs_minimize_permol_kernels.cu:796-932): a call
 * is split by size — one, two or four waves per system with the vectors in LDS, and from 656 coordinates on (option
 * NVMK_BFGS_TEAM) a TEAM of 2 ... 32 workgroups per system that deal the force field's terms among themselves and keep the
 * inverse Hessian as the history of its rank-2 updates (the (xi, H dGrad) pairs, dealt over the team: where 3 max_iters <=
 * 2 x coordinates; option NVMK_BFGS_HISTORY=0: the packed triangle, rows dealt over the team — the same H_k in exact
 * arithmetic, other roundings); a system's results depend on its size, max_iters and the options only (bitwise reproducible
 * for every class up to 9000 coordinates).
 * The workgroups of a team wait for each other inside the launch: calls with team systems take turns per device inside the
 * library, and a team whose members do not all become resident (another PROCESS holding the device's CUs with a kernel that
 * never ends) gives up after NVMK_BFGS_TEAM_TIMEOUT_MS and the call returns NVMK_ERR_INVALID_ARGUMENT with that message. */
int nvmk_bfgs_minimize(const nvmk_ff_batch* batch, const int32_t* h_atom_starts, double w0, double w1, int max_iters,
                       double grad_tol, int scale_grads, double* d_pos, const uint8_t* d_active, double* d_energies,
                       int16_t* d_statuses, int32_t* d_iters, void* stream);
/* The same with repeatUntilConverged inside the launch (reference: etkdg_stage_distgeom_minimize.cu:53-58 relaunches the
 * minimiser while any system is unconverged): a system that stops at max_iters is minimised again from where it stands, with
 * a fresh inverse Hessian, up to `restarts` more times.  Results are those of restarts + 1 calls of nvmk_bfgs_minimize on
 * the still-unconverged systems; d_iters reports the last minimisation's iterations. */
int nvmk_bfgs_minimize_repeat(const nvmk_ff_batch* batch, const int32_t* h_atom_starts, double w0, double w1, int max_iters,
                              int restarts, double grad_tol, int scale_grads, double* d_pos, const uint8_t* d_active,
                              double* d_energies, int16_t* d_statuses, int32_t* d_iters, void* stream);

/* Two minimisations of every system in ONE launch, each with its own weights, iteration limit and restarts — the first
 * distance-geometry minimisation of ETKDG (chiral 1.0, 4th dimension 0.1, 400 iterations) and its fourth-dimension stage
 * (0.2, 1.0, 200; reference: src/etkdg_stage_distgeom_minimize.cu:177-249, two DistGeomMinimizeStage objects with the
 * stereo checks of etkdg_stage_stereochem_checks.cu in between).  The coordinates after the first minimisation are left in
 * d_pos_between for the checks that sit between the stages; d_pos, energies, statuses and iterations report the second.  A
 * system whose first-stage energy exceeds skip_above_energy_per_atom x atoms (< 0: never) is left as the first stage left it:
 * the caller is going to fail it on that energy anyway.  Each system's results are those of two separate calls. */
typedef struct nvmk_bfgs_second_stage {
  double  w0, w1;
  int32_t max_iters, restarts;
  double* d_pos_between;
  double  skip_above_energy_per_atom;
} nvmk_bfgs_second_stage;
int nvmk_bfgs_minimize_two_stages(const nvmk_ff_batch* batch, const int32_t* h_atom_starts, double w0, double w1, int max_iters,
                                  int restarts, const nvmk_bfgs_second_stage* second /* NULL: one stage */, double grad_tol,
                                  int scale_grads, double* d_pos, const uint8_t* d_active, double* d_energies,
                                  int16_t* d_statuses, int32_t* d_iters, void* stream);

/* Measurement hook (bench.py's roofline, tests): when d_counters != NULL every BFGS launch of this process — the ones
 * nvmk_etkdg_embed issues included — adds to d_counters[8 * kind + k] (device memory, 64 uint64, caller zeroes it):
 * k = 0 systems minimised, 1 BFGS iterations, 2 inverse-Hessian bytes those iterations stand for (read + write of the
 * packed triangle, 8 n (n + 2) per iteration: SURVEY.md 8(d)'s algorithmic bytes), 3 energy evaluations, 4 the part of
 * (2) whose rows were not resident in LDS, i.e. the bytes actually requested from HBM.  kind as nvmk_ff_batch.kind
 * (constraint variants of MMFF / UFF count as 5 / 6).  NULL switches the counters off (default).  The reference has no
 * counterpart; its benchmarks time whole calls (benchmarks/bench_utils/timing.py:53-91). */
int nvmk_bfgs_set_stats(uint64_t* d_counters);

/* ---- E1: ETKDG attempt scheduler -------------------------------------------------------------------------
 * Replaces nvMolKit::detail::Scheduler (src/etkdg_impl.h:223-280, src/etkdg_impl.cpp:272-326): round-robin dispatch
 * of molecule ids, at most confs_per_mol * max_iterations attempts per molecule, oversubscription once every
 * molecule has had confs_per_mol attempts.  Thread-safe.  create() returns NULL (and sets the error slot) unless all
 * three parameters are > 0.  dispatch() writes up to batch_size ids; record() takes -1 for a failed attempt. */
void* nvmk_scheduler_create(int n_mols, int confs_per_mol, int max_iterations);
void  nvmk_scheduler_destroy(void* scheduler);
int   nvmk_scheduler_dispatch(void* scheduler, int batch_size, int32_t* h_mol_ids_out, int* n_out);
int   nvmk_scheduler_record(void* scheduler, const int32_t* h_mol_ids, const int16_t* h_finished_on_iteration, int n);

/* ---- E2-E8: batched ETKDG embedding on flattened molecules -----------------------------------------------
 * Replaces nvMolKit::embedMolecules (src/etkdg.cpp:90-484) downstream of RDKit: the caller supplies, per UNIQUE
 * molecule, the DG and ETK term groups (layouts of nvmk_ff_batch, `starts` indexed by molecule) and the list of
 * stereochemistry checks that RDKit's EmbedArgs hold (tetrahedralCarbons, chiralCenters, doubleBondEnds,
 * stereoDoubleBonds; src/embedder_utils.cpp:229-347).  Stage order, weights and thresholds are the reference's
 * (src/etkdg.cpp:331-419): random 4-D coordinates -> DG minimise (1.0, 0.1, 400 iters, E/atom < 0.05) -> tetrahedral
 * check -> [first chiral check] -> DG minimise (0.2, 1.0, 200) -> [ETK minimise 300 + planarity] -> double-bond
 * geometry -> [final chiral volume, chiral distances, chiral centre-in-volume, double-bond stereo].
 *
 * Stereo check term = kind + 5 local atom indices + 2 doubles:
 *   NVMK_CHECK_TETRAHEDRAL           idx(centre, n1, n2, n3, n4 or centre)  par(inFusedSmallRings, -)
 *   NVMK_CHECK_CHIRAL_VOLUME         idx(-, 1, 2, 3, 4)                     par(volLower, volUpper)
 *   NVMK_CHECK_CHIRAL_DISTANCE       idx(i, j, -, -, -)                     par(lower, upper)
 *   NVMK_CHECK_CHIRAL_CENTER_VOLUME  idx as TETRAHEDRAL                     -
 *   NVMK_CHECK_DOUBLE_BOND_STEREO    idx(0, 1, 2, 3, -)                     par(sign, -)
 *   NVMK_CHECK_DOUBLE_BOND_GEOMETRY  idx(0, 1, 2, -, -)                     -
 * Surplus attempts (option NVMK_ETKDG_PRUNE, default 1): a molecule that misses k conformers when a batch starts sends only k + 1
 * of its surviving attempts past stage 4 (ETK minimisation onwards); the others leave without a failure being counted, but they DO
 * count against the molecule's attempt budget (confs_per_mol x max_iterations), so with a very small max_iterations a molecule
 * whose kept attempts fail late can end with fewer conformers than the unpruned pipeline, and WHICH attempts are accepted
 * depends on the molecule's count at the start of the batch.  NVMK_ETKDG_PRUNE=0 runs every attempt through every stage (the
 * reference's behaviour).
 * Output: conformer c of molecule m starts at d_coords[3 * (confs_per_mol * sum_{k<m} n_atoms[k] + c * n_atoms[m])],
 * h_conf_counts[m] conformers are valid.  h_stage_failures (optional, NVMK_ETKDG_N_STAGES ints) totals failures per
 * stage (the reference's ETKDGContext::totalFailures).  Blocking. */
#define NVMK_CHECK_TETRAHEDRAL 0
#define NVMK_CHECK_CHIRAL_VOLUME 1
#define NVMK_CHECK_CHIRAL_DISTANCE 2
#define NVMK_CHECK_CHIRAL_CENTER_VOLUME 3
#define NVMK_CHECK_DOUBLE_BOND_STEREO 4
#define NVMK_CHECK_DOUBLE_BOND_GEOMETRY 5
#define NVMK_ETKDG_N_STAGES 11

typedef struct nvmk_etkdg_molset {
  int32_t        n_mols;
  const int32_t* h_n_atoms;        /* HOST [n_mols] */
  nvmk_ff_group  dg[3];            /* DEVICE, starts [n_mols + 1] */
  nvmk_ff_group  etk[6];           /* DEVICE, starts [n_mols + 1]; may be all-NULL when the ETK stage is off */
  const int32_t* check_starts;     /* DEVICE [n_mols + 1], NULL = no checks */
  const int32_t* check_kind;
  const int32_t* check_idx;        /* 5 per term */
  const double*  check_par;        /* 2 per term */
  const int32_t* num_impropers;    /* DEVICE [n_mols] (planarity tolerance 0.7 * num_impropers) */
  const int32_t* h_etk_d12_counts; /* HOST [n_mols]: terms of etk[2] / etk[3] per molecule */
  const int32_t* h_etk_d13_counts;
  const void*    build_handle;     /* NULL, or the nvmk_etkdg_molset_build handle these tables belong to (nvmk_etkdg_molset_view sets
                                      it): nvmk_etkdg_embed then meets the rows batch by batch — before a batch runs it waits
                                      (nvmk_etkdg_molset_wait) until the rows of the batch's molecules have been uploaded, so a build
                                      with NVMK_BUILD_ASYNC fills the tables of batch k + 1 while batch k is on the GPU (the
                                      reference: per-batch host flattening on the batch's OpenMP thread, src/etkdg.cpp:175-191,211-240) */
} nvmk_etkdg_molset;

typedef struct nvmk_etkdg_params {
  int32_t  confs_per_mol;
  int32_t  max_iterations;     /* attempts per conformer (reference: 10 x atoms when -1, src/etkdg.cpp:71-85,195-197) */
  int32_t  batch_size;         /* conformer attempts per batch (reference default 500) */
  int32_t  use_exp_torsions;   /* EmbedParameters::useExpTorsionAnglePrefs */
  int32_t  use_basic_knowledge;
  int32_t  enforce_chirality;
  double   box_size;           /* 5 * boxSizeMult, or -boxSizeMult if negative (etkdg_stage_coordgen.cu:101-106) */
  double   force_tol;          /* EmbedParameters::optimizerForceTol */
  uint64_t seed;
  int32_t  batches_per_gpu;    /* concurrent batches (worker threads + streams), BatchHardwareOptions::batchesPerGpu
                                  (src/hardware_options.h:26-35); <= 1: one batch at a time, bit-reproducible for a
                                  seed; > 1: faster, but which attempt a molecule gets depends on batch timing */
} nvmk_etkdg_params;

int nvmk_etkdg_embed(const nvmk_etkdg_molset* mols, const nvmk_etkdg_params* params, double* d_coords,
                     int32_t* h_conf_counts, int32_t* h_stage_failures, void* stream);

/* ---- host-side table assembly: per-molecule host arrays -> the resident, device-ordered tables above ----------------------
 * Replaces the reference's per-batch host preprocessing between RDKit and its kernels: the flatteners' "add to batch" step
 * (addMoleculeToBatch / addMoleculeToMolecularSystem, src/forcefields/dist_geom.h / mmff.h; callers src/etkdg.cpp:175-191,
 * :211-240 on `preprocessingThreads` OpenMP threads, src/minimizer/bfgs_mmff.cpp:139-201), which its benchmarks time as part
 * of EmbedMolecules / MMFFOptimizeMoleculesConfs (benchmarks/etkdg_bench.py:108-124).
 *
 * Input: one descriptor per molecule pointing at the caller's own per-molecule term arrays (nvmk_host_terms: n_terms rows of
 * n_idx LOCAL atom indices — int32, or int64 as numpy makes them by default — and n_par doubles, C order; layouts as
 * nvmk_ff_batch).  Work: n_threads host threads (<= 0: all, at most 64; the reference's preprocessingThreads) concatenate
 * the groups molecule by molecule, convert the indices, bring the O(N^2) pair groups into the order the kernels want (along
 * the diagonals of the pair matrix: (|j - i|, min(i, j)), stable — 64 consecutive terms then touch 128 distinct atoms instead
 * of sharing atom i; DG g0, ETK g5, MMFF g5 / g6, UFF g4), and for MMFF merge van der Waals and electrostatic pairs into
 * group 11; they write into a ring of pinned staging slots owned by the library, from which the chunks go to ONE device
 * allocation with hipMemcpyAsync on `stream` while the threads fill the next slots (build of chunk k + 1 overlaps the upload
 * of chunk k).  The uploads run on a stream of the build's own, forked from `stream`.  Blocking on the host work unless
 * NVMK_BUILD_ASYNC is set; on return the last uploads are still in flight and `stream` has been made to wait for them: work
 * enqueued on `stream` afterwards sees complete tables, a consumer on ANOTHER stream calls nvmk_*_wait(handle, ..., its stream)
 * first (host wait until the rows have been handed to the copy engine + the stream waits for those copies).  The handle owns the
 * device memory; nvmk_*_view fills the plain structs the entry points above take (valid until nvmk_*_free, which waits for the
 * uploads).
 *   NVMK_BUILD_ASYNC           : return once the plan is made (sizes known, device block allocated, views valid): host threads of
 *                                the build's own fill and upload the rows in molecule order while the caller goes on.  The
 *                                caller's arrays must stay alive until nvmk_*_wait(handle, -1, ...) or nvmk_*_free has returned;
 *                                EVERY consumer waits first — nvmk_etkdg_embed does so itself, batch by batch, through
 *                                nvmk_etkdg_molset.build_handle.  Errors of the fill are reported by the wait.
 *   NVMK_BUILD_KEEP_PAIR_ORDER : pair groups keep the caller's row order (measurements)
 *   NVMK_BUILD_NO_MMFF_MERGE   : no group 11
 *   NVMK_BUILD_HOST            : the tables are written to HOST memory instead (no GPU involved; the views then hold host
 *                                pointers) — how the CPU test-suite checks the assembly row by row
 * MMFF group 11 exists only if EVERY molecule's electrostatic pairs are a subset of its van der Waals pairs and no pair is
 * listed twice (how RDKit and the reference's builder emit them); otherwise the view's groups[11] stays NULL. */
typedef struct nvmk_host_terms {
  int32_t       n_terms;
  int32_t       idx_bytes; /* 4: idx is int32_t[], 8: int64_t[]; ignored when n_terms == 0 */
  const void*   idx;       /* n_terms x n_idx */
  const double* par;       /* n_terms x n_par (ignored for groups without parameters) */
} nvmk_host_terms;

typedef struct nvmk_flat_molecule {
  int32_t         n_atoms;
  int32_t         num_impropers; /* planarity tolerance of the basic-knowledge check = 0.7 x this */
  int32_t         has_etk;       /* 0: etk[] is ignored (all molecules of a set must agree) */
  int32_t         n_checks;
  nvmk_host_terms dg[3];
  nvmk_host_terms etk[6];
  const int32_t*  check_kind;    /* [n_checks] NVMK_CHECK_* */
  const int32_t*  check_idx;     /* [n_checks][5] */
  const double*   check_par;     /* [n_checks][2] */
} nvmk_flat_molecule;

#define NVMK_BUILD_KEEP_PAIR_ORDER 1u
#define NVMK_BUILD_NO_MMFF_MERGE 2u
#define NVMK_BUILD_HOST 4u
#define NVMK_BUILD_ASYNC 8u

int nvmk_etkdg_molset_build(const nvmk_flat_molecule* h_mols, int32_t n_mols, int n_threads, unsigned flags, void* stream,
                            void** handle);
int nvmk_etkdg_molset_view(const void* handle, nvmk_etkdg_molset* out);
/* rows of molecules [0, first_n_mols) (negative: all) uploaded before what `stream` runs next; blocks the host while they are being filled */
int nvmk_etkdg_molset_wait(const void* handle, int32_t first_n_mols, void* stream);
int nvmk_etkdg_molset_free(void* handle);
/* Per-MOLECULE term tables of one force field (kind NVMK_FF_DG / ETK / MMFF / UFF) for nvmk_ff_batch.system_mol batches:
 * h_terms[m * n_groups + g] are molecule m's rows of group g; n_groups = the kind's group count (3 / 6 / 7 / 5), for MMFF and
 * UFF optionally followed by up to four constraint groups (distance, position, angle, torsion).  The view fills
 * groups[0 .. n_groups) and, for MMFF, groups[11]; every `starts` array has n_mols + 1 entries.  A "molecule" here is whatever
 * shares one copy of the tables: with one row per SYSTEM and no system_mol the same call assembles a plain batch. */
int nvmk_ff_tables_build(int kind, const nvmk_host_terms* h_terms, int32_t n_mols, int n_groups, int n_threads, unsigned flags,
                         void* stream, void** handle);
int nvmk_ff_tables_view(const void* handle, nvmk_ff_group groups[12], int32_t* n_mols);
int nvmk_ff_tables_wait(const void* handle, void* stream); /* as nvmk_etkdg_molset_wait, every molecule */
int nvmk_ff_tables_free(void* handle);

/* Per-stage wall-clock table of the LAST nvmk_etkdg_embed call of the process that ran with the option NVMK_ETKDG_TIMING=1
 * (reference: ETKDGDriver's debug mode, src/etkdg_impl.cpp:126-139 recordStageTiming, :161-200 printTimingStatistics: total /
 * min / max / calls per stage).  Rows 0 .. NVMK_ETKDG_N_STAGES - 1 are the stages (one entry per batch), row NVMK_ETKDG_N_STAGES
 * the host work of a batch outside its stages (scheduler dispatch, uploads, record, pack), row NVMK_ETKDG_N_STAGES + 1 the
 * whole call.  With the option set every stage ends with a stream synchronisation (what makes the clock meaningful), which
 * costs throughput; without it nothing is recorded.  `names` (optional) receives pointers to static strings — the stage
 * names of the reference's pipeline.  Every entry point, ETKDG stage and BFGS size class additionally opens a roctx range
 * (reference: ScopedNvtxRange, src/utils/nvtx.h:36-69) — `rocprofv3 --marker-trace`; option NVMK_MARKERS=0 turns them off. */
#define NVMK_ETKDG_TIMING_ROWS (NVMK_ETKDG_N_STAGES + 2)
int nvmk_etkdg_stage_timings(double* total_ms, double* min_ms, double* max_ms, int32_t* calls, int n_rows, const char** names);

/* Units of the pipeline above exposed on their own so that E2 / E3 can be tested the way the reference tests them.
 *
 * nvmk_etkdg_random_coords: stage 0 (reference: ETKDGCoordGenStage, src/etkdg_stage_coordgen.cu:83-127): every active
 * system s gets 4 * n_atoms uniform coordinates in [-box_size / 2, box_size / 2), drawn from the counter-based generator
 * keyed by (seed, attempt_base + s, coordinate index) — the reference draws them from RDKit's RNG on the host.
 *
 * nvmk_etkdg_driver_run: the driver's bookkeeping (reference: ETKDGDriver::run / iterate, src/etkdg_impl.cpp:111-149,
 * kernels src/etkdg_kernels.cu:20-70) with programmed stages: h_failed[(stage * max_iterations + iteration) * n_systems
 * + s] != 0 fails system s there.  Outputs (host): h_fail_counts[stage * n_systems + s], h_finished_on[s] (-1 = never),
 * the number of finished systems and the iterations run.  Errors like the reference's constructor: no systems, no
 * stages. */
int nvmk_etkdg_random_coords(uint64_t seed, uint64_t attempt_base, int n_systems, const int32_t* d_atom_starts,
                             const uint8_t* d_active, double box_size, double* d_pos, void* stream);
int nvmk_etkdg_driver_run(int n_systems, int n_stages, int max_iterations, const uint8_t* h_failed, int16_t* h_fail_counts,
                          int16_t* h_finished_on, int32_t* h_n_finished, int32_t* h_iterations, void* stream);

/* One stereochemistry check stage on given coordinates: d_failed[s] is set to 1 for every active system s on which a
 * term of `kind` fails (never cleared).  Replaces the execute() of the reference's check stages
 * (ETKDGTetrahedralCheckStage, ETKDGFirstChiralCenterCheckStage, ETKDGChiralCenterVolumeCheckStage,
 * ETKDGChiralDistMatrixCheckStage, ETKDGDoubleBondStereoCheckStage, ETKDGDoubleBondGeometryCheckStage:
 * src/etkdg_stage_stereochem_checks.cu:25-442, .h:69,122) as a unit of its own, so that E6 can be tested directly; the
 * embedding pipeline above runs the same kernel.  d_pos holds 4 doubles per atom (x, y, z, w; w ignored), atom_starts has
 * n_systems + 1 entries, sys_mol[s] selects the molecule whose check terms (nvmk_etkdg_molset layout) apply to system s,
 * d_active may be NULL (all active). */
int nvmk_etkdg_stereo_check(int kind, int n_systems, const int32_t* d_atom_starts, const int32_t* d_sys_mol,
                            const int32_t* d_check_starts, const int32_t* d_check_kind, const int32_t* d_check_idx,
                            const double* d_check_par, const double* d_pos, const uint8_t* d_active, uint8_t* d_failed,
                            void* stream);

/* ---- conformer RMSD matrices and RMS pruning (the step right after the embedding) ---------
 * Replaces conformerRmsdBatchMatrixGpu (src/conformer_rmsd.h:58-85, src/conformer_rmsd.cu:262-392) and the CPU pruning loop
 * of addConformersToMoleculeWithPruning (rdkit_extensions/conformer_pruning.cpp:88-137).
 * Molecule m has n_confs[m] conformers of n_atoms[m] atoms at d_coords[coord_offsets[m] + (conf * n_atoms[m] + atom) * 3 + xyz];
 * its n (n - 1) / 2 pair values start at d_out[pair_offsets[m]] in condensed lower-triangle order: pair (i, j), i > j, at
 * i (i - 1) / 2 + j (the order of RDKit's GetConformerRMSMatrix).  prealigned != 0: plain RMSD of the raw coordinates;
 * otherwise each pair is optimally superimposed first (Kabsch, proper rotations only).  All pointers are device pointers;
 * pair_offsets has n_mols + 1 entries with pair_offsets[n_mols] == total_pairs.
 */
int nvmk_conformer_rmsd_batch(const double* d_coords, const int64_t* d_coord_offsets, const int32_t* d_n_atoms,
                              const int64_t* d_pair_offsets, int n_mols, int64_t total_pairs, int prealigned, double* d_out,
                              void* stream);

/* The same with the molecule's symmetry: every molecule m brings K_m atom mappings of L_m atoms each (d_matches, mapping k of
 * molecule m at d_match_offsets[m] + k * d_match_len[m]; K_m = (d_match_offsets[m + 1] - d_match_offsets[m]) / d_match_len[m],
 * at least one) and the entry of a pair is the SMALLEST optimally superposed RMSD over the mappings — conformer i's atoms of
 * mapping 0 against conformer j's atoms of mapping k (reference: _isConfFarFromRest, conformer_pruning.cpp:88-114, which loops
 * over the self matches and rejects a conformer as soon as one of them brings it within the threshold of a kept one).  With one
 * mapping per molecule this is the RMSD over an atom subset (onlyHeavyAtomsForRMS).  Condensed pair order as above. */
int nvmk_conformer_rmsd_batch_sym(const double* d_coords, const int64_t* d_coord_offsets, const int32_t* d_n_atoms,
                                  const int64_t* d_pair_offsets, int n_mols, int64_t total_pairs, const int64_t* d_match_offsets,
                                  const int32_t* d_match_len, const int32_t* d_matches, double* d_out, void* stream);
/* Greedy pruning on the matrices above: conformer i of a molecule is kept (d_keep[conf_starts[m] + i] = 1) iff its RMSD
 * to every conformer kept before it is >= threshold.  conf_starts has n_mols + 1 entries. */
int nvmk_conformer_prune(const double* d_rmsd, const int64_t* d_pair_offsets, const int32_t* d_conf_starts, int n_mols,
                         double threshold, uint8_t* d_keep, void* stream);

/* ---- SMILES ingestion for the fingerprint path (host code, no GPU involved) ----------------
 * Replaces what the reference takes from RDKit before MorganInvariantsGenerator::ComputeInvariantsInto can run
 * (RDKit::SmilesToMol + sanitisation: src/morgan_fingerprint_common.cpp:43-124 works on ROMol objects; the reference's
 * benchmarks read benchmarks/data/chembl_10k.smi through RDKit).  Scope and rules: nvmolkit_amd/csrc/smiles.cpp — like RDKit's
 * sanitisation every molecule is Kekulised and its aromaticity perceived again (RDKit's default model); by default the result
 * must equal what the input wrote (true for SMILES written by RDKit), else the molecule is REFUSED per molecule, as are
 * valences RDKit's sanitisation rejects (the spellings its cleanUp step rewrites, e.g. N(=O)=O, are rewritten alike):
 * nothing is ever fingerprinted differently from RDKit.
 *   nvmk_smiles_parse        : n_mols NUL- or whitespace-terminated strings -> an opaque set of graphs (n_threads <= 0: all
 *                              host threads).  Never fails on bad chemistry: the per-molecule status says what happened.
 *   nvmk_smiles_parse_text   : the same for one text buffer with a molecule per line (a .smi file as it is read from disk:
 *                              the SMILES is the first blank-separated column, "\n" or "\r\n" line ends; every line counts,
 *                              an empty one is an empty molecule; the buffer need not end in a newline or NUL) - no
 *                              per-molecule strings on the caller's side.  nvmk_smiles_size says how many molecules there were.
 *   nvmk_smiles_counts       : atoms / bonds (after folding [H] atoms) and status of every molecule; any output may be NULL
 *   nvmk_smiles_graph        : one molecule's graph, for tests and other consumers: atom_fields[6 * n_atoms] =
 *                              (Z, formal charge, isotope, total H count, aromatic, in ring), bond_fields[4 * n_bonds] =
 *                              (begin, end, RDKit bond type 1 / 2 / 3 / 4 / 12, in ring)
 *   nvmk_smiles_morgan_inputs: the five HOST arrays nvmk_morgan_from_invariants consumes, for the selected molecules
 *                              (mol_ids NULL = all, in order) in slots of max_atoms; every selected molecule must have
 *                              status NVMK_SMILES_OK and atoms, bonds < max_atoms (the reference's bucketing rule).
 */
#define NVMK_SMILES_OK 0
#define NVMK_SMILES_SYNTAX_ERROR 1
#define NVMK_SMILES_VALENCE_ERROR 2      /* RDKit: "Explicit valence for atom ... is greater than permitted" */
#define NVMK_SMILES_NEEDS_AROMATICITY 3  /* RDKit would perceive the aromaticity differently from what the input wrote (Kekule form) */
#define NVMK_SMILES_TOO_MANY_BONDS 4     /* more than 8 bonds on one atom (kMaxBondsPerAtom of the reference) */
#define NVMK_SMILES_NO_KEKULE_FORM 5     /* RDKit: "Can't kekulize mol" / "non-ring atom marked aromatic" */
#define NVMK_SMILES_UNSUPPORTED_ISOTOPE 6 /* isotope label outside the library's mass table whose mass defect could change int(mass - weight) */
int nvmk_smiles_parse(const char* const* smiles, int64_t n_mols, int n_threads, void** handle);
/* flags = NVMK_SMILES_PERCEIVE_AROMATICITY: the perceived aromaticity (RDKit's default model: electron donation rules and
 * fused-ring combinations; checked against the aromaticity RDKit recorded in the reference's ChEMBL SMILES: all 8864 aromatic
 * molecules of the 10 000 reproduced) is applied, whatever form the input was written in, instead of refusing a difference. */
#define NVMK_SMILES_PERCEIVE_AROMATICITY 1u
int nvmk_smiles_parse_flags(const char* const* smiles, int64_t n_mols, int n_threads, unsigned flags, void** handle);
int nvmk_smiles_parse_text(const char* text, int64_t n_bytes, int n_threads, unsigned flags, void** handle);
/* The same set of graphs from the MDL molfile (V2000) records of an SD file held in memory ("$$$$"-separated; data items are
 * skipped): what RDKit's SDMolSupplier (sanitize, removeHs) would hand to the fingerprint generator.  Every record counts;
 * V3000 and query atoms / bonds give NVMK_SMILES_SYNTAX_ERROR.  All nvmk_smiles_* accessors apply to the handle. */
int nvmk_sdf_parse_text(const char* text, int64_t n_bytes, int n_threads, unsigned flags, void** handle);
int nvmk_smiles_size(const void* handle, int64_t* n_mols);
int nvmk_smiles_free(void* handle);
int nvmk_smiles_counts(const void* handle, int32_t* n_atoms, int32_t* n_bonds, int8_t* status);
int nvmk_smiles_graph(const void* handle, int64_t mol, int32_t* atom_fields, int32_t* bond_fields);
/* Self-matches of one molecule's (hydrogen-free) graph for symmetry-aware RMS pruning — what the reference takes from RDKit:
 * SubstructMatch(mol, mol, maxMatches = 1000, uniquify = false) after removeHs, optionally on the copy whose conjugated
 * terminal groups were made symmetric (rdkit_extensions/conformer_pruning.cpp:24-60 getMolSelfMatches;
 * EmbedParameters::symmetrizeConjugatedTerminalGroupsForPruning).  out[k * n_atoms + i] = image of atom i in match k, the
 * identity first; at most max_matches matches are written (out must hold max_matches * n_atoms ints), *n_matches says how many.
 * Returns NVMK_TRUNCATED (> 0, the matches written are valid) when the backtracking search gave up on its step budget of 5e7
 * before it had found max_matches matches or exhausted the mappings: the list is then NOT the whole group. */
int nvmk_smiles_self_matches(const void* handle, int64_t mol, int symmetrize_terminal, int max_matches, int32_t* out,
                             int32_t* n_matches);
int nvmk_smiles_morgan_inputs(const void* handle, const int64_t* mol_ids, int64_t n_sel, int max_atoms, uint32_t* atom_inv,
                              uint32_t* bond_inv, int16_t* bond_idx, int16_t* bond_other, int16_t* n_atoms, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* NVMOLKIT_AMD_H */
