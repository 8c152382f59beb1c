// Host-side table assembly: per-molecule term arrays -> the resident, kernel-ordered tables of a molecule set, on host
// threads, through a ring of pinned staging slots, uploaded chunk by chunk while the next chunks are being filled.
//
// Reference counterpart: the per-batch host preprocessing between RDKit's contribs and the kernels — addMoleculeToBatch /
// addMoleculeToMolecularSystem (src/forcefields/dist_geom.h:367-410, mmff.h:370, uff.h:216) called from OpenMP regions of
// `preprocessingThreads` threads (src/etkdg.cpp:175-191, :211-240; src/minimizer/bfgs_mmff.cpp:139-213) followed by
// AsyncDeviceVector copies.  The reference appends every CONFORMER's terms to host vectors and uploads each batch whole; here
// a molecule's terms are stored once (conformers share them through nvmk_ff_batch.system_mol), every row's final position
// is known from a prefix sum before any row is written, so the threads write rows straight into pinned memory in their final
// order and the copy engine starts on chunk 0 while chunk 1 is being filled.
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <chrono>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "common.h"
#include "options.h"

namespace nvmk {
namespace tables {
namespace {

constexpr size_t kSlotBytes  = 32u << 20;  // pinned staging: kSlots x kSlotBytes per ring (a ring grows to its largest molecule)
constexpr int    kSlots      = 4;
constexpr int    kMaxThreads = 64;
constexpr size_t kAlign      = 256;
inline size_t    align_bytes(const size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

struct GroupShape {
  int  nIdx, nPar;
  bool pairOrder;
};
// (n_idx, n_par) of every term group per kind (include/nvmolkit_amd.h); pairOrder marks the O(N^2) tables
const GroupShape kDg[3]   = {{2, 3, true}, {4, 2, false}, {1, 0, false}};
const GroupShape kEtk[6]  = {{4, 12, false}, {4, 4, false}, {2, 4, false}, {2, 4, false}, {3, 2, false}, {2, 4, true}};
const GroupShape kMmff[7] = {{2, 2, false}, {3, 3, false}, {3, 5, false}, {4, 1, false}, {4, 3, false}, {2, 2, true}, {2, 3, true}};
const GroupShape kUff[5]  = {{2, 2, false}, {3, 6, false}, {4, 3, false}, {4, 4, false}, {2, 3, true}};
// optional constraint groups behind the MMFF / UFF groups: distance, position, angle, torsion
const GroupShape kConstraint[4] = {{2, 3, false}, {1, 5, false}, {3, 3, false}, {4, 3, false}};

const GroupShape* shapes_of(const int kind, int* n) {
  switch (kind) {
    case NVMK_FF_DG: *n = 3; return kDg;
    case NVMK_FF_ETK: *n = 6; return kEtk;
    case NVMK_FF_MMFF: *n = 7; return kMmff;
    case NVMK_FF_UFF: *n = 5; return kUff;
    default: *n = 0; return nullptr;
  }
}

// ---- pinned staging rings, reused across calls (pinning pages costs ~0.3 s per GB: never per call) ------------------------
struct Ring {
  char*      base      = nullptr;
  size_t     slotBytes = 0;
  hipEvent_t lastUse   = nullptr;  // recorded after the last upload out of this ring; the next owner waits for it
  bool       busy      = false;
};
std::mutex                         g_ringMutex;
std::vector<std::unique_ptr<Ring>> g_rings;

int acquire_ring(const size_t minSlotBytes, Ring** out) {
  Ring* r = nullptr;
  {
    const std::lock_guard<std::mutex> lock(g_ringMutex);
    for (auto& c : g_rings) {
      if (!c->busy) {
        r = c.get();
        break;
      }
    }
    if (r == nullptr) {
      g_rings.emplace_back(new Ring());
      r = g_rings.back().get();
    }
    r->busy = true;
  }
  if (r->lastUse != nullptr) {  // the previous owner's last uploads out of these slots (an event of ITS device and stream)
    (void)hipEventSynchronize(r->lastUse);
    (void)hipEventDestroy(r->lastUse);
    r->lastUse = nullptr;
  }
  if (r->slotBytes < minSlotBytes) {
    if (r->base != nullptr) (void)hipHostFree(r->base);
    r->base      = nullptr;
    r->slotBytes = 0;
    void*            p = nullptr;
    const hipError_t e = hipHostMalloc(&p, minSlotBytes * kSlots, hipHostMallocPortable);
    if (e != hipSuccess) {
      const std::lock_guard<std::mutex> lock(g_ringMutex);
      r->busy = false;
      set_last_error("table build: hipHostMalloc of %zu staging bytes failed: %s", minSlotBytes * kSlots, hipGetErrorString(e));
      return NVMK_ERR_OUT_OF_MEMORY;
    }
    r->base      = static_cast<char*>(p);
    r->slotBytes = minSlotBytes;
  }
  *out = r;
  return NVMK_OK;
}

// `recordUse`: uploads out of the ring may still be in flight on `stream`.  The event is made HERE, under the device that is
// current for this build (rings are shared by every device of the process, events are not); if it cannot be recorded the stream
// is drained instead — the next owner must never write into a slot that is still being read.
void release_ring(Ring* r, hipStream_t stream, const bool recordUse) {
  if (r == nullptr) return;
  if (recordUse) {
    hipEvent_t ev = nullptr;
    hipError_t e  = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ev, stream);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      if (ev != nullptr) (void)hipEventDestroy(ev);
      ev = nullptr;
      (void)hipStreamSynchronize(stream);
    }
    r->lastUse = ev;
  }
  const std::lock_guard<std::mutex> lock(g_ringMutex);
  r->busy = false;
}

// ---- copy streams: one per build in flight, leased from a pool that lives for the process ------------------------------------
// (never destroyed: the staging rings keep an event recorded on the stream of their last user, and an event must not outlive the
// stream it was recorded on — a copy stream that died with its build left the next owner of the ring synchronising on such an
// event: hipErrorCapturedEvent out of nowhere, some hundred builds later)
struct CopyStream {
  hipStream_t stream = nullptr;
  int         device = -1;
  bool        busy   = false;
};
std::vector<CopyStream> g_copyStreams;  // guarded by g_ringMutex

int lease_copy_stream(const int device, hipStream_t* out) {
  {
    const std::lock_guard<std::mutex> lock(g_ringMutex);
    for (CopyStream& c : g_copyStreams) {
      if (!c.busy && c.device == device) {
        c.busy = true;
        *out   = c.stream;
        return NVMK_OK;
      }
    }
  }
  hipStream_t s = nullptr;
  NVMK_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const std::lock_guard<std::mutex> lock(g_ringMutex);
  g_copyStreams.push_back({s, device, true});
  *out = s;
  return NVMK_OK;
}
void return_copy_stream(hipStream_t s) {
  if (s == nullptr) return;
  const std::lock_guard<std::mutex> lock(g_ringMutex);
  for (CopyStream& c : g_copyStreams)
    if (c.stream == s) c.busy = false;
}

// the ring of one build: given back on every way out of run()
struct RingLease {
  Ring*       ring      = nullptr;
  hipStream_t stream    = nullptr;
  bool        recordUse = true;  // (uploads may have been issued on any path that got as far as holding a ring)
  ~RingLease() { release_ring(ring, stream, recordUse); }
};

// ---- the plan: groups, their source descriptors, where every row goes -----------------------------------------------------
enum Fill { kPlain, kPairOrdered, kMergedNonbonded };

struct Group {
  int  nIdx = 0, nPar = 0;
  Fill fill = kPlain;
  // source rows of molecule m: src[m * srcStride]; kMergedNonbonded: src = van der Waals rows, src2 = electrostatic rows
  const char* src       = nullptr;
  const char* src2      = nullptr;
  size_t      srcStride = 0;  // bytes between the descriptors of consecutive molecules
  bool        checkIdx  = true;
  std::vector<int32_t> starts;                                        // [n_mols + 1]
  size_t               startsOff = 0, idxOff = 0, parOff = 0;         // byte offsets in the destination block
  size_t               idx_row() const { return static_cast<size_t>(nIdx) * 4; }
  size_t               par_row() const { return static_cast<size_t>(nPar) * 8; }
  const nvmk_host_terms& terms(const int m) const { return *reinterpret_cast<const nvmk_host_terms*>(src + static_cast<size_t>(m) * srcStride); }
  const nvmk_host_terms& terms2(const int m) const { return *reinterpret_cast<const nvmk_host_terms*>(src2 + static_cast<size_t>(m) * srcStride); }
};

inline int64_t read_idx(const nvmk_host_terms& t, const size_t k) {
  return t.idx_bytes == 8 ? static_cast<const int64_t*>(t.idx)[k] : static_cast<const int32_t*>(t.idx)[k];
}

struct PairRow {
  uint32_t lo, hi;
  int32_t  row, ele;
};
struct Scratch {
  std::vector<uint64_t> keys;  // (sort key << 24) | source row
  std::vector<PairRow>  pairs;
};
constexpr int kMaxPairRows = 1 << 24;

// Rows of molecule m of group g, written to dstIdx / dstPar in their final order.  Returns false on an atom index outside
// [0, nAtoms) (nAtoms < 0: not checked).
bool fill_rows(const Group& g, const int m, const int nAtoms, const unsigned flags, int32_t* dstIdx, double* dstPar, Scratch& sc,
               std::atomic<int>& mergeImpossible) {
  const nvmk_host_terms& t = g.terms(m);
  const int              n = t.n_terms;
  if (n <= 0) return true;
  const bool keepOrder = (flags & NVMK_BUILD_KEEP_PAIR_ORDER) != 0;
  const bool checked   = g.checkIdx && nAtoms >= 0;
  bool       ok        = true;
  auto       take      = [&](const int64_t v) {
    if (checked && (v < 0 || v >= nAtoms)) ok = false;
    return static_cast<int32_t>(v);
  };
  if (g.fill == kPlain || (g.fill == kPairOrdered && keepOrder)) {
    const size_t cells = static_cast<size_t>(n) * g.nIdx;
    if (t.idx_bytes == 4 && !checked) {
      std::memcpy(dstIdx, t.idx, cells * 4);
    } else {
      for (size_t k = 0; k < cells; ++k) dstIdx[k] = take(read_idx(t, k));
    }
    if (g.nPar > 0) std::memcpy(dstPar, t.par, static_cast<size_t>(n) * g.par_row());
    return ok;
  }
  if (n >= kMaxPairRows) return false;
  if (g.fill == kPairOrdered) {  // by (|j - i|, min(i, j)), rows with equal keys in the caller's order
    sc.keys.resize(static_cast<size_t>(n));
    for (int r = 0; r < n; ++r) {
      const int64_t  a = take(read_idx(t, 2 * static_cast<size_t>(r))), b = take(read_idx(t, 2 * static_cast<size_t>(r) + 1));
      const uint64_t lo = static_cast<uint64_t>(std::min(a, b)) & 0xfffff, d = static_cast<uint64_t>(std::max(a, b) - std::min(a, b)) & 0xfffff;
      sc.keys[static_cast<size_t>(r)] = (((d << 20) | lo) << 24) | static_cast<uint64_t>(r);
    }
    std::sort(sc.keys.begin(), sc.keys.end());
    for (int r = 0; r < n; ++r) {
      const size_t s    = static_cast<size_t>(sc.keys[static_cast<size_t>(r)] & 0xffffff);
      dstIdx[2 * r]     = static_cast<int32_t>(read_idx(t, 2 * s));
      dstIdx[2 * r + 1] = static_cast<int32_t>(read_idx(t, 2 * s + 1));
      if (g.nPar > 0) std::memcpy(dstPar + static_cast<size_t>(r) * g.nPar, t.par + s * g.nPar, g.par_row());
    }
    return ok;
  }
  // kMergedNonbonded: one row per van der Waals pair — (R*, eps) of that row, (chargeTerm, dielModel, is1_4) of the electrostatic
  // row of the same pair or zeros.  Impossible (no group 11 for the whole set) when a pair is listed twice or an electrostatic
  // pair has no van der Waals row.
  const nvmk_host_terms& e = g.terms2(m);
  if (e.n_terms >= kMaxPairRows) return false;
  sc.pairs.resize(static_cast<size_t>(n));
  for (int r = 0; r < n; ++r) {
    const int64_t a = take(read_idx(t, 2 * static_cast<size_t>(r))), b = take(read_idx(t, 2 * static_cast<size_t>(r) + 1));
    sc.pairs[static_cast<size_t>(r)] = {static_cast<uint32_t>(std::min(a, b)), static_cast<uint32_t>(std::max(a, b)), r, -1};
  }
  const auto byPair = [](const PairRow& x, const PairRow& y) { return x.lo != y.lo ? x.lo < y.lo : x.hi < y.hi; };
  std::sort(sc.pairs.begin(), sc.pairs.end(), byPair);
  for (int r = 1; r < n; ++r) {
    if (sc.pairs[static_cast<size_t>(r)].lo == sc.pairs[static_cast<size_t>(r) - 1].lo &&
        sc.pairs[static_cast<size_t>(r)].hi == sc.pairs[static_cast<size_t>(r) - 1].hi) {
      mergeImpossible.store(1);
      return ok;
    }
  }
  for (int r = 0; r < e.n_terms; ++r) {
    const int64_t a = read_idx(e, 2 * static_cast<size_t>(r)), b = read_idx(e, 2 * static_cast<size_t>(r) + 1);
    const PairRow key{static_cast<uint32_t>(std::min(a, b)), static_cast<uint32_t>(std::max(a, b)), 0, 0};
    const auto    it = std::lower_bound(sc.pairs.begin(), sc.pairs.end(), key, byPair);
    if (it == sc.pairs.end() || it->lo != key.lo || it->hi != key.hi || it->ele >= 0) {
      mergeImpossible.store(1);
      return ok;
    }
    it->ele = r;
  }
  if (!keepOrder) {  // along the diagonals; the pairs are unique, so the keys alone decide
    std::sort(sc.pairs.begin(), sc.pairs.end(), [](const PairRow& x, const PairRow& y) {
      const uint32_t dx = x.hi - x.lo, dy = y.hi - y.lo;
      return dx != dy ? dx < dy : x.lo < y.lo;
    });
  }
  for (int r = 0; r < n; ++r) {
    const PairRow& p  = sc.pairs[static_cast<size_t>(r)];
    const size_t   s  = static_cast<size_t>(p.row);
    dstIdx[2 * r]     = static_cast<int32_t>(read_idx(t, 2 * s));
    dstIdx[2 * r + 1] = static_cast<int32_t>(read_idx(t, 2 * s + 1));
    double* q         = dstPar + static_cast<size_t>(r) * 5;
    q[0]              = t.par[2 * s];
    q[1]              = t.par[2 * s + 1];
    if (p.ele >= 0) {
      const double* ep = e.par + static_cast<size_t>(p.ele) * 3;
      q[2] = ep[0], q[3] = ep[1], q[4] = ep[2];
    } else {
      q[2] = q[3] = q[4] = 0.0;
    }
  }
  return ok;
}

// ---- one build: plan, destination block, the chunked fill + upload ------------------------------------------------------
struct Build {
  int                  nMols = 0;
  unsigned             flags = 0;
  std::vector<Group>   groups;
  const int32_t*       nAtoms = nullptr;  // per molecule, or NULL (indices not range-checked)
  size_t               nAtomsStride = 0;  // bytes
  // small per-molecule int32 columns that travel in the header next to the `starts` arrays
  std::vector<std::vector<int32_t>> extraColumns;
  std::vector<size_t>               extraOff;
  std::vector<char>    header;      // all starts arrays + extra columns, as they lie at the front of the block
  size_t               blockBytes = 0;
  char*                block      = nullptr;  // device (hipMalloc) or host (malloc) destination
  bool                 onHost     = false;
  int                  device     = -1;
  hipStream_t          stream     = nullptr;  // the caller's
  hipStream_t          copyStream = nullptr;  // the uploads run on a stream of the build's own, forked from the caller's (ADVICE r05:
                                              // slot recycling then waits for the chunk's copy, not for whatever else is queued)
  hipEvent_t           done       = nullptr;  // after the last upload
  hipEvent_t           fork       = nullptr;  // on the caller's stream when the build began: the copy stream waits for it
  std::atomic<int>     mergeImpossible{0};
  int                  mergedGroup = -1;
  // The plan of the fill (run_plan) and, with NVMK_BUILD_ASYNC, the thread that carries it out while the caller goes on: the rows
  // of the first molsReady molecules have been handed to the copy engine, chunkEvent[c] follows chunk c's copies.
  std::vector<int>        chunkFirst;      // first molecule of every chunk, then nMols
  std::vector<int>        chunkOf;         // per molecule
  std::vector<hipEvent_t> chunkEvent;      // per chunk (device builds)
  std::vector<std::vector<size_t>> chunkSlotOff;  // per chunk, per group: idx offset, par offset inside the slot
  size_t                  slotBytes = 0;
  RingLease               lease;
  std::thread             builder;
  std::atomic<int>        molsReady{0};
  std::atomic<int>        finished{0};
  int                     asyncRc = NVMK_OK;
  std::string             asyncError;

  int n_atoms_of(const int m) const {
    return nAtoms == nullptr ? -1 : *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(nAtoms) + static_cast<size_t>(m) * nAtomsStride);
  }
  ~Build() {
    if (builder.joinable()) builder.join();  // (reads the caller's arrays until it is through)
    if (done != nullptr) {
      (void)hipEventSynchronize(done);
      (void)hipEventDestroy(done);
    } else if (copyStream != nullptr) {
      (void)hipStreamSynchronize(copyStream);
    }
    for (hipEvent_t ev : chunkEvent)
      if (ev != nullptr) (void)hipEventDestroy(ev);
    if (fork != nullptr) (void)hipEventDestroy(fork);
    lease.stream = copyStream;
    release_ring(lease.ring, copyStream, lease.ring != nullptr);  // (before the stream goes: the ring's last-use event is recorded on it)
    lease.ring = nullptr;
    return_copy_stream(copyStream);  // (drained above: `done` follows its last copy, or the stream was synchronised)
    if (block != nullptr) {
      if (onHost) {
        std::free(block);
      } else {
        (void)hipFree(block);
      }
    }
  }
};

int validate_terms(const nvmk_host_terms& t, const int nPar, const char* what, const int m, const int g) {
  NVMK_REQUIRE(t.n_terms >= 0, "%s: molecule %d group %d: negative term count", what, m, g);
  if (t.n_terms == 0) return NVMK_OK;
  NVMK_REQUIRE(t.idx != nullptr && (t.idx_bytes == 4 || t.idx_bytes == 8), "%s: molecule %d group %d: idx must be int32 or int64 rows", what, m, g);
  NVMK_REQUIRE(nPar == 0 || t.par != nullptr, "%s: molecule %d group %d: par is NULL", what, m, g);
  return NVMK_OK;
}

int run_plan(Build& b, const char* what) {
  const int nMols = b.nMols;
  // 1. starts of every group (and the total size of a molecule's rows, for the chunking)
  std::vector<size_t> molBytes(static_cast<size_t>(nMols), 0);
  for (Group& g : b.groups) {
    g.starts.assign(static_cast<size_t>(nMols) + 1, 0);
    const size_t rowBytes = g.idx_row() + g.par_row();
    int64_t      total    = 0;
    for (int m = 0; m < nMols; ++m) {
      total += g.terms(m).n_terms;
      NVMK_REQUIRE(total < (1ll << 31), "%s: more than 2^31 rows in one term group", what);
      g.starts[static_cast<size_t>(m) + 1] = static_cast<int32_t>(total);
      molBytes[static_cast<size_t>(m)] += align_bytes(static_cast<size_t>(g.terms(m).n_terms) * rowBytes);
    }
  }
  // 2. layout of the destination block: [starts of every group | extra columns] [idx g0] [par g0] ...
  size_t off = 0;
  for (Group& g : b.groups) {
    g.startsOff = off;
    off += (static_cast<size_t>(nMols) + 1) * 4;
    off = (off + kAlign - 1) / kAlign * kAlign;
  }
  b.extraOff.clear();
  for (const auto& col : b.extraColumns) {
    b.extraOff.push_back(off);
    off += col.size() * 4;
    off = (off + kAlign - 1) / kAlign * kAlign;
  }
  const size_t headerBytes = off;
  for (Group& g : b.groups) {
    g.idxOff = off;
    off += static_cast<size_t>(g.starts.back()) * g.idx_row();
    off = (off + kAlign - 1) / kAlign * kAlign;
    g.parOff = off;
    off += static_cast<size_t>(g.starts.back()) * g.par_row();
    off = (off + kAlign - 1) / kAlign * kAlign;
  }
  b.blockBytes = std::max<size_t>(off, kAlign);
  b.header.assign(headerBytes, 0);
  for (const Group& g : b.groups) std::memcpy(b.header.data() + g.startsOff, g.starts.data(), g.starts.size() * 4);
  for (size_t c = 0; c < b.extraColumns.size(); ++c)
    std::memcpy(b.header.data() + b.extraOff[c], b.extraColumns[c].data(), b.extraColumns[c].size() * 4);

  // 3. chunks of molecules that fit a staging slot
  size_t largest = 0;
  for (const size_t x : molBytes) largest = std::max(largest, x);
  const size_t nGroups2  = b.groups.size() * 2;
  const long   slotKb    = opt::get(opt::kBuildSlotKb).num(0);
  const size_t slotBytes = std::max(slotKb > 0 ? static_cast<size_t>(slotKb) << 10 : kSlotBytes, largest + nGroups2 * kAlign);
  b.slotBytes            = slotBytes;
  std::vector<int>& chunkFirst = b.chunkFirst;
  chunkFirst.assign(1, 0);
  {
    size_t used = nGroups2 * kAlign;
    for (int m = 0; m < nMols; ++m) {
      if (used + molBytes[static_cast<size_t>(m)] > slotBytes && m > chunkFirst.back()) {
        chunkFirst.push_back(m);
        used = nGroups2 * kAlign;
      }
      used += molBytes[static_cast<size_t>(m)];
    }
    chunkFirst.push_back(nMols);
  }
  const int nChunks = static_cast<int>(chunkFirst.size()) - 1;
  b.chunkOf.assign(static_cast<size_t>(nMols), 0);
  b.chunkSlotOff.assign(static_cast<size_t>(nChunks), {});
  for (int c = 0; c < nChunks; ++c) {
    std::vector<size_t>& so = b.chunkSlotOff[static_cast<size_t>(c)];
    so.resize(nGroups2);
    size_t    o  = 0;
    const int m0 = chunkFirst[static_cast<size_t>(c)], m1 = chunkFirst[static_cast<size_t>(c) + 1];
    for (size_t gi = 0; gi < b.groups.size(); ++gi) {
      const Group& g    = b.groups[gi];
      const size_t rows = static_cast<size_t>(g.starts[static_cast<size_t>(m1)] - g.starts[static_cast<size_t>(m0)]);
      so[2 * gi]        = o;
      o                 = (o + rows * g.idx_row() + kAlign - 1) / kAlign * kAlign;
      so[2 * gi + 1]    = o;
      o                 = (o + rows * g.par_row() + kAlign - 1) / kAlign * kAlign;
    }
    NVMK_REQUIRE(b.onHost || o <= slotBytes, "%s: internal: chunk %d does not fit its staging slot", what, c);
    for (int m = m0; m < m1; ++m) b.chunkOf[static_cast<size_t>(m)] = c;
  }

  // 4. destination; the uploads' own stream, forked from the caller's (what the caller queued before this call precedes them)
  if (b.onHost) {
    b.block = static_cast<char*>(std::malloc(b.blockBytes));
    NVMK_REQUIRE(b.block != nullptr, "%s: out of host memory (%zu bytes)", what, b.blockBytes);
    std::memcpy(b.block, b.header.data(), headerBytes);
  } else {
    NVMK_HIP_CHECK(hipGetDevice(&b.device));
    void* p = nullptr;
    NVMK_HIP_CHECK(hipMalloc(&p, b.blockBytes));
    b.block = static_cast<char*>(p);
    if (const int rcs = lease_copy_stream(b.device, &b.copyStream)) return rcs;
    NVMK_HIP_CHECK(hipEventCreateWithFlags(&b.fork, hipEventDisableTiming));
    NVMK_HIP_CHECK(hipEventRecord(b.fork, b.stream));
    NVMK_HIP_CHECK(hipStreamWaitEvent(b.copyStream, b.fork, 0));
    NVMK_HIP_CHECK(hipMemcpyAsync(b.block, b.header.data(), headerBytes, hipMemcpyHostToDevice, b.copyStream));  // b.header lives as long as the handle
    b.chunkEvent.assign(static_cast<size_t>(nChunks), nullptr);
    for (int c = 0; c < nChunks; ++c) NVMK_HIP_CHECK(hipEventCreateWithFlags(&b.chunkEvent[static_cast<size_t>(c)], hipEventDisableTiming));
    NVMK_HIP_CHECK(hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
    b.lease.stream = b.copyStream;
    const int rc   = acquire_ring(slotBytes, &b.lease.ring);
    if (rc != NVMK_OK) return rc;
  }
  return NVMK_OK;
}

// 5. fill (worker threads) and upload (the calling thread), chunk by chunk.  Workers take molecules in order from one counter; a
// molecule of chunk c may be written once slot c % kSlots is free again (chunkOpen > c).  After chunk c's copies have been handed
// to the copy engine its event is recorded and molsReady moves past its molecules: nvmk_*_wait builds on both.
// A thread that has nothing to do yet (its molecule's staging slot is still being uploaded; the chunk it is to upload is still being
// filled) gives the core away: a few yields, then it sleeps.  Spinning on yield alone, the 60 of 64 workers that wait for one of the
// four slots kept every core they could get busy for the whole fill — the whole ChEMBL file's molecule set took 2.54 s to assemble on
// 64 threads against 0.95 s on 16, its MMFF tables 4.39 s against 1.07 s, and the ETKDG batches running beside them 18.5 s against
// 16.4 s (profiles/r06_conformers/table_assembly_by_threads.txt).
inline void idle(const int spins) {
  if (spins < 16) {
    std::this_thread::yield();
  } else {
    std::this_thread::sleep_for(std::chrono::microseconds(spins < 64 ? 20 : 100));
  }
}

int run_fill(Build& b, const int nThreadsAsked, const char* what) {
  const int               nMols      = b.nMols;
  const std::vector<int>& chunkFirst = b.chunkFirst;
  const int               nChunks    = static_cast<int>(chunkFirst.size()) - 1;
  Ring* const             ring       = b.lease.ring;
  std::vector<std::atomic<int>> remaining(static_cast<size_t>(nChunks));
  for (int c = 0; c < nChunks; ++c) remaining[static_cast<size_t>(c)].store(chunkFirst[static_cast<size_t>(c) + 1] - chunkFirst[static_cast<size_t>(c)]);
  std::atomic<int> nextMol{0}, chunkOpen{b.onHost ? nChunks : std::min(nChunks, kSlots)}, badMol{-1}, badGroup{-1}, abort{0}, threw{0};
  auto worker_body = [&]() {
    Scratch sc;
    for (;;) {
      const int m = nextMol.fetch_add(1);
      if (m >= nMols) return;
      const int c = b.chunkOf[static_cast<size_t>(m)];
      for (int spins = 0; chunkOpen.load(std::memory_order_acquire) <= c; ++spins) {
        if (abort.load()) return;
        idle(spins);
      }
      const std::vector<size_t>& so    = b.chunkSlotOff[static_cast<size_t>(c)];
      const int                  m0    = chunkFirst[static_cast<size_t>(c)];
      const int                  atoms = b.n_atoms_of(m);
      for (size_t gi = 0; gi < b.groups.size(); ++gi) {
        const Group& g = b.groups[gi];
        int32_t*     dIdx;
        double*      dPar;
        if (b.onHost) {
          dIdx = reinterpret_cast<int32_t*>(b.block + g.idxOff + static_cast<size_t>(g.starts[static_cast<size_t>(m)]) * g.idx_row());
          dPar = reinterpret_cast<double*>(b.block + g.parOff + static_cast<size_t>(g.starts[static_cast<size_t>(m)]) * g.par_row());
        } else {
          char*        slot = ring->base + static_cast<size_t>(c % kSlots) * ring->slotBytes;
          const size_t r0   = static_cast<size_t>(g.starts[static_cast<size_t>(m)] - g.starts[static_cast<size_t>(m0)]);
          dIdx = reinterpret_cast<int32_t*>(slot + so[2 * gi] + r0 * g.idx_row());
          dPar = reinterpret_cast<double*>(slot + so[2 * gi + 1] + r0 * g.par_row());
        }
        if (!fill_rows(g, m, atoms, b.flags, dIdx, dPar, sc, b.mergeImpossible)) {
          int expected = -1;
          if (badMol.compare_exchange_strong(expected, m)) badGroup.store(static_cast<int>(gi));
        }
      }
      remaining[static_cast<size_t>(c)].fetch_sub(1, std::memory_order_release);
    }
  };
  // (a worker that runs out of memory in its sort scratch must not take the process down: the build fails instead)
  auto worker = [&]() {
    try {
      worker_body();
    } catch (...) {
      threw.store(1);
      abort.store(1);
    }
  };
  int nThreads = nThreadsAsked > 0 ? nThreadsAsked : static_cast<int>(std::thread::hardware_concurrency());
  nThreads     = std::max(1, std::min({nThreads, kMaxThreads, nMols / 8 + 1}));
  std::vector<std::thread> pool;
  pool.reserve(static_cast<size_t>(nThreads));
  for (int t = 0; t < nThreads; ++t) {
    try {
      pool.emplace_back(worker);
    } catch (const std::system_error&) {  // the host refuses more threads: go on with those it gave
      break;
    }
  }
  NVMK_REQUIRE(!pool.empty(), "%s: could not start a worker thread", what);
  int rc = NVMK_OK;
  if (!b.onHost) {
    for (int c = 0; c < nChunks && rc == NVMK_OK; ++c) {
      for (int spins = 0; remaining[static_cast<size_t>(c)].load(std::memory_order_acquire) > 0 && !abort.load(); ++spins) idle(spins);
      if (abort.load()) break;
      const std::vector<size_t>& so = b.chunkSlotOff[static_cast<size_t>(c)];
      char*     slot = ring->base + static_cast<size_t>(c % kSlots) * ring->slotBytes;
      const int m0 = chunkFirst[static_cast<size_t>(c)], m1 = chunkFirst[static_cast<size_t>(c) + 1];
      for (size_t gi = 0; gi < b.groups.size() && rc == NVMK_OK; ++gi) {
        const Group& g    = b.groups[gi];
        const size_t r0   = static_cast<size_t>(g.starts[static_cast<size_t>(m0)]);
        const size_t rows = static_cast<size_t>(g.starts[static_cast<size_t>(m1)]) - r0;
        if (rows == 0) continue;
        hipError_t e = hipMemcpyAsync(b.block + g.idxOff + r0 * g.idx_row(), slot + so[2 * gi], rows * g.idx_row(), hipMemcpyHostToDevice, b.copyStream);
        if (e == hipSuccess && g.nPar > 0)
          e = hipMemcpyAsync(b.block + g.parOff + r0 * g.par_row(), slot + so[2 * gi + 1], rows * g.par_row(), hipMemcpyHostToDevice, b.copyStream);
        if (e != hipSuccess) {
          set_last_error("%s: upload of chunk %d failed: %s", what, c, hipGetErrorString(e));
          rc = NVMK_ERR_HIP;
        }
      }
      if (rc == NVMK_OK) {
        hipError_t e = hipEventRecord(b.chunkEvent[static_cast<size_t>(c)], b.copyStream);
        if (e == hipSuccess) b.molsReady.store(m1, std::memory_order_release);
        // the slot is needed again: open chunk c + kSlots once this upload is through (its own copies, nothing else is on this stream)
        if (e == hipSuccess && c + kSlots < nChunks) e = hipEventSynchronize(b.chunkEvent[static_cast<size_t>(c)]);
        if (e != hipSuccess) {
          set_last_error("%s: waiting for the upload of chunk %d failed: %s", what, c, hipGetErrorString(e));
          rc = NVMK_ERR_HIP;
        }
      }
      chunkOpen.store(std::min(nChunks, c + kSlots + 1), std::memory_order_release);
    }
    if (rc != NVMK_OK) abort.store(1);
  }
  for (auto& t : pool) t.join();
  if (threw.load() && rc == NVMK_OK) {
    set_last_error("%s: a worker thread ran out of memory", what);
    rc = NVMK_ERR_OUT_OF_MEMORY;
  }
  if (!b.onHost && rc == NVMK_OK) {
    const hipError_t e = hipEventRecord(b.done, b.copyStream);
    if (e != hipSuccess) {
      set_last_error("%s: recording the completion event failed: %s", what, hipGetErrorString(e));
      rc = NVMK_ERR_HIP;
    }
  }
  if (!b.onHost) {  // the staging ring goes back to the pool (its last uploads may still be in flight: the next owner waits for them)
    release_ring(b.lease.ring, b.copyStream, true);
    b.lease.ring = nullptr;
  }
  if (rc != NVMK_OK) return rc;
  NVMK_REQUIRE(badMol.load() < 0, "%s: molecule %d, term group %d: atom index outside the molecule (or more than 2^24 pair rows)", what,
               badMol.load(), badGroup.load());
  b.molsReady.store(nMols, std::memory_order_release);
  return NVMK_OK;
}

// Plan, then fill: at once (the caller's stream then waits for the last upload: what it queues afterwards sees complete tables), or
// — NVMK_BUILD_ASYNC — on a thread of the build's own while the caller goes on; nvmk_*_wait is then how a consumer meets the rows.
int run(Build& b, const int nThreadsAsked, const char* what) {
  if (const int rc = run_plan(b, what)) return rc;
  if ((b.flags & NVMK_BUILD_ASYNC) != 0 && !b.onHost) {
    Build* self = &b;
    b.builder   = std::thread([self, nThreadsAsked, what]() {
      int rc = NVMK_ERR_INTERNAL;
      try {
        rc = hipSetDevice(self->device) == hipSuccess ? run_fill(*self, nThreadsAsked, what) : NVMK_ERR_HIP;
      } catch (...) {
        set_last_error("%s: the builder thread failed", what);
      }
      if (rc != NVMK_OK) self->asyncError = nvmk_last_error();  // (this thread's message)
      self->asyncRc = rc;
      self->finished.store(1, std::memory_order_release);
    });
    return NVMK_OK;
  }
  const int rc = run_fill(b, nThreadsAsked, what);
  b.asyncRc    = rc;
  b.finished.store(1, std::memory_order_release);
  if (rc == NVMK_OK && !b.onHost) NVMK_HIP_CHECK(hipStreamWaitEvent(b.stream, b.done, 0));
  return rc;
}

// A consumer on `stream` needs the rows of molecules [0, firstN) (negative: all): wait on the host until their copies have been
// handed to the copy engine, then let `stream` wait for those copies.
int build_wait(Build& b, int firstN, hipStream_t stream, const char* what) {
  firstN = firstN < 0 ? b.nMols : std::min(firstN, b.nMols);
  while (b.molsReady.load(std::memory_order_acquire) < firstN && b.finished.load(std::memory_order_acquire) == 0) {
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  if (b.finished.load(std::memory_order_acquire) != 0 && b.asyncRc != NVMK_OK) {
    set_last_error("%s: the table build failed: %s", what, b.asyncError.empty() ? "(no message)" : b.asyncError.c_str());
    return b.asyncRc;
  }
  if (b.onHost || firstN <= 0) return NVMK_OK;
  if (b.molsReady.load(std::memory_order_acquire) >= b.nMols && b.finished.load(std::memory_order_acquire) != 0) {
    NVMK_HIP_CHECK(hipStreamWaitEvent(stream, b.done, 0));
  } else {
    NVMK_HIP_CHECK(hipStreamWaitEvent(stream, b.chunkEvent[static_cast<size_t>(b.chunkOf[static_cast<size_t>(firstN) - 1])], 0));
  }
  return NVMK_OK;
}

void view_group(const Build& b, const Group& g, nvmk_ff_group* out) {
  out->starts = reinterpret_cast<const int32_t*>(b.block + g.startsOff);
  out->idx    = g.starts.back() > 0 ? reinterpret_cast<const int32_t*>(b.block + g.idxOff) : nullptr;
  out->par    = g.starts.back() > 0 && g.nPar > 0 ? reinterpret_cast<const double*>(b.block + g.parOff) : nullptr;
}

void add_groups(Build& b, const GroupShape* shapes, const int n, const char* src, const size_t stride) {
  for (int g = 0; g < n; ++g) {
    Group x;
    x.nIdx      = shapes[g].nIdx;
    x.nPar      = shapes[g].nPar;
    x.fill      = shapes[g].pairOrder ? kPairOrdered : kPlain;
    x.src       = src + static_cast<size_t>(g) * sizeof(nvmk_host_terms);
    x.srcStride = stride;
    b.groups.push_back(std::move(x));
  }
}

// (`build` LAST: members die in reverse order, and an asynchronous build's thread — joined by ~Build — reads the vectors before it)
struct MolsetHandle {
  uint32_t             magic = 0x4d4f4c53;  // "MOLS"
  bool                 hasEtk = false, hasChecks = false;
  std::vector<int32_t> nAtoms, d12, d13;
  std::vector<nvmk_host_terms> checkTerms;  // 2 per molecule: (check_idx, check_par) and (check_kind, -)
  Build                build;
};
struct TablesHandle {
  uint32_t magic = 0x5441424c;  // "TABL"
  int      kind = 0, nGroups = 0;
  Build    build;
};

}  // namespace

}  // namespace tables
}  // namespace nvmk

using namespace nvmk::tables;

namespace {

// The C ABI does not let C++ exceptions out: a build that runs out of host memory while planning (vectors of the size of the
// molecule set) is an error code like any other.
template <typename F> int no_throw(const char* what, F&& body) {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    nvmk::set_last_error("%s: out of host memory", what);
    return NVMK_ERR_OUT_OF_MEMORY;
  } catch (const std::exception& e) {
    nvmk::set_last_error("%s: %s", what, e.what());
    return NVMK_ERR_INTERNAL;
  }
}

int molset_build(const nvmk_flat_molecule* h_mols, int32_t n_mols, int n_threads, unsigned flags, void* stream, void** handle) {
  NVMK_REQUIRE(n_mols >= 0 && (n_mols == 0 || h_mols != nullptr), "nvmk_etkdg_molset_build: bad molecule array");
  std::unique_ptr<MolsetHandle> h(new MolsetHandle());
  Build&                        b = h->build;
  b.nMols  = n_mols;
  b.flags  = flags;
  b.onHost = (flags & NVMK_BUILD_HOST) != 0;
  b.stream = nvmk::as_stream(stream);
  h->hasEtk = n_mols > 0;
  int64_t nChecks = 0;
  h->nAtoms.resize(static_cast<size_t>(n_mols));
  std::vector<int32_t> impropers(static_cast<size_t>(n_mols));
  for (int m = 0; m < n_mols; ++m) {
    const nvmk_flat_molecule& mol = h_mols[m];
    NVMK_REQUIRE(mol.n_atoms >= 0, "nvmk_etkdg_molset_build: molecule %d: negative atom count", m);
    NVMK_REQUIRE(mol.n_checks >= 0 && (mol.n_checks == 0 || (mol.check_kind && mol.check_idx && mol.check_par)),
                 "nvmk_etkdg_molset_build: molecule %d: bad stereo-check arrays", m);
    for (int g = 0; g < 3; ++g)
      if (const int rc = validate_terms(mol.dg[g], kDg[g].nPar, "nvmk_etkdg_molset_build (dg)", m, g)) return rc;
    if (mol.has_etk) {
      for (int g = 0; g < 6; ++g)
        if (const int rc = validate_terms(mol.etk[g], kEtk[g].nPar, "nvmk_etkdg_molset_build (etk)", m, g)) return rc;
    } else {
      h->hasEtk = false;
    }
    h->nAtoms[static_cast<size_t>(m)]   = mol.n_atoms;
    impropers[static_cast<size_t>(m)]   = mol.num_impropers;
    nChecks += mol.n_checks;
  }
  const char* base = reinterpret_cast<const char*>(h_mols);
  add_groups(b, kDg, 3, base + offsetof(nvmk_flat_molecule, dg), sizeof(nvmk_flat_molecule));
  if (h->hasEtk) {
    add_groups(b, kEtk, 6, base + offsetof(nvmk_flat_molecule, etk), sizeof(nvmk_flat_molecule));
    h->d12.resize(static_cast<size_t>(n_mols));
    h->d13.resize(static_cast<size_t>(n_mols));
    for (int m = 0; m < n_mols; ++m) {
      h->d12[static_cast<size_t>(m)] = h_mols[m].etk[2].n_terms;
      h->d13[static_cast<size_t>(m)] = h_mols[m].etk[3].n_terms;
    }
  }
  h->hasChecks = nChecks > 0;
  if (h->hasChecks) {  // the checks as two more groups with the same row counts: (idx[5], par[2]) and (kind)
    h->checkTerms.resize(static_cast<size_t>(n_mols) * 2);
    for (int m = 0; m < n_mols; ++m) {
      h->checkTerms[2 * static_cast<size_t>(m)]     = {h_mols[m].n_checks, 4, h_mols[m].check_idx, h_mols[m].check_par};
      h->checkTerms[2 * static_cast<size_t>(m) + 1] = {h_mols[m].n_checks, 4, h_mols[m].check_kind, nullptr};
    }
    static const GroupShape kCheck[2] = {{5, 2, false}, {1, 0, false}};
    add_groups(b, kCheck, 2, reinterpret_cast<const char*>(h->checkTerms.data()), 2 * sizeof(nvmk_host_terms));
    b.groups[b.groups.size() - 1].checkIdx = false;
    b.groups[b.groups.size() - 2].checkIdx = false;  // unused slots of a check's five indices may hold anything
  }
  b.extraColumns.push_back(std::move(impropers));
  b.nAtoms       = h->nAtoms.data();
  b.nAtomsStride = sizeof(int32_t);
  if (const int rc = run(b, n_threads, "nvmk_etkdg_molset_build")) return rc;
  *handle = h.release();
  return NVMK_OK;
}

}  // namespace

extern "C" {

int nvmk_etkdg_molset_build(const nvmk_flat_molecule* h_mols, int32_t n_mols, int n_threads, unsigned flags, void* stream, void** handle) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(handle != nullptr, "nvmk_etkdg_molset_build: handle is NULL");
  *handle = nullptr;
  return no_throw("nvmk_etkdg_molset_build", [&]() { return molset_build(h_mols, n_mols, n_threads, flags, stream, handle); });
}

int nvmk_etkdg_molset_view(const void* handle, nvmk_etkdg_molset* out) {
  const MolsetHandle* h = static_cast<const MolsetHandle*>(handle);
  NVMK_REQUIRE(h != nullptr && h->magic == 0x4d4f4c53 && out != nullptr, "nvmk_etkdg_molset_view: not a molecule-set handle");
  std::memset(out, 0, sizeof(*out));
  const Build& b = h->build;
  out->n_mols    = b.nMols;
  out->h_n_atoms = h->nAtoms.data();
  size_t gi = 0;
  for (int g = 0; g < 3; ++g) view_group(b, b.groups[gi++], &out->dg[g]);
  if (h->hasEtk) {
    for (int g = 0; g < 6; ++g) view_group(b, b.groups[gi++], &out->etk[g]);
    out->h_etk_d12_counts = h->d12.data();
    out->h_etk_d13_counts = h->d13.data();
  }
  if (h->hasChecks) {
    nvmk_ff_group a{}, k{};
    view_group(b, b.groups[gi], &a);
    view_group(b, b.groups[gi + 1], &k);
    out->check_starts = a.starts;
    out->check_idx    = a.idx;
    out->check_par    = a.par;
    out->check_kind   = k.idx;
  }
  out->num_impropers = reinterpret_cast<const int32_t*>(b.block + b.extraOff[0]);
  out->build_handle  = b.onHost ? nullptr : handle;  // nvmk_etkdg_embed meets the rows batch by batch (nvmk_etkdg_molset_wait)
  return NVMK_OK;
}

int nvmk_etkdg_molset_wait(const void* handle, int32_t first_n_mols, void* stream) {
  MolsetHandle* h = static_cast<MolsetHandle*>(const_cast<void*>(handle));
  NVMK_REQUIRE(h != nullptr && h->magic == 0x4d4f4c53, "nvmk_etkdg_molset_wait: not a molecule-set handle");
  return build_wait(h->build, first_n_mols, nvmk::as_stream(stream), "nvmk_etkdg_molset_wait");
}

int nvmk_etkdg_molset_free(void* handle) {
  if (handle == nullptr) return NVMK_OK;
  MolsetHandle* h = static_cast<MolsetHandle*>(handle);
  NVMK_REQUIRE(h->magic == 0x4d4f4c53, "nvmk_etkdg_molset_free: not a molecule-set handle");
  h->magic = 0;
  if (h->build.builder.joinable()) h->build.builder.join();  // an asynchronous fill reads the handle's arrays (and the caller's) to its end
  delete h;
  return NVMK_OK;
}

}  // extern "C"

namespace {

int tables_build(int kind, const nvmk_host_terms* h_terms, int32_t n_mols, int n_groups, int n_threads, unsigned flags, void* stream,
                 void** handle) {
  int               nShapes = 0;
  const GroupShape* shapes  = shapes_of(kind, &nShapes);
  NVMK_REQUIRE(shapes != nullptr, "nvmk_ff_tables_build: kind must be NVMK_FF_DG, _ETK, _MMFF or _UFF");
  const int maxExtra = (kind == NVMK_FF_MMFF || kind == NVMK_FF_UFF) ? 4 : 0;
  NVMK_REQUIRE(n_groups >= nShapes && n_groups <= nShapes + maxExtra, "nvmk_ff_tables_build: kind %d has %d term groups%s, got %d", kind, nShapes,
               maxExtra ? " (+ up to 4 constraint groups)" : "", n_groups);
  NVMK_REQUIRE(n_mols >= 0 && (n_mols == 0 || h_terms != nullptr), "nvmk_ff_tables_build: bad term array");
  GroupShape all[12];
  for (int g = 0; g < n_groups; ++g) all[g] = g < nShapes ? shapes[g] : kConstraint[g - nShapes];
  shapes = all;
  for (int m = 0; m < n_mols; ++m)
    for (int g = 0; g < n_groups; ++g)
      if (const int rc = validate_terms(h_terms[static_cast<size_t>(m) * n_groups + g], shapes[g].nPar, "nvmk_ff_tables_build", m, g)) return rc;
  std::unique_ptr<TablesHandle> h(new TablesHandle());
  h->kind    = kind;
  h->nGroups = n_groups;
  Build& b   = h->build;
  b.nMols    = n_mols;
  b.flags    = flags;
  b.onHost   = (flags & NVMK_BUILD_HOST) != 0;
  b.stream   = nvmk::as_stream(stream);
  const size_t stride = static_cast<size_t>(n_groups) * sizeof(nvmk_host_terms);
  add_groups(b, shapes, n_groups, reinterpret_cast<const char*>(h_terms), stride);
  int64_t vdwRows = 0;
  if (kind == NVMK_FF_MMFF)
    for (int m = 0; m < n_mols; ++m) vdwRows += h_terms[static_cast<size_t>(m) * n_groups + 5].n_terms;
  if (kind == NVMK_FF_MMFF && (flags & NVMK_BUILD_NO_MMFF_MERGE) == 0 && vdwRows > 0) {
    Group x;
    x.nIdx      = 2;
    x.nPar      = 5;
    x.fill      = kMergedNonbonded;
    x.src       = reinterpret_cast<const char*>(h_terms + 5);
    x.src2      = reinterpret_cast<const char*>(h_terms + 6);
    x.srcStride = stride;
    b.mergedGroup = static_cast<int>(b.groups.size());
    b.groups.push_back(std::move(x));
  }
  if (const int rc = run(b, n_threads, "nvmk_ff_tables_build")) return rc;
  *handle = h.release();
  return NVMK_OK;
}

}  // namespace

extern "C" {

int nvmk_ff_tables_build(int kind, const nvmk_host_terms* h_terms, int32_t n_mols, int n_groups, int n_threads, unsigned flags, void* stream,
                         void** handle) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(handle != nullptr, "nvmk_ff_tables_build: handle is NULL");
  *handle = nullptr;
  return no_throw("nvmk_ff_tables_build", [&]() { return tables_build(kind, h_terms, n_mols, n_groups, n_threads, flags, stream, handle); });
}

int nvmk_ff_tables_view(const void* handle, nvmk_ff_group groups[12], int32_t* n_mols) {
  const TablesHandle* h = static_cast<const TablesHandle*>(handle);
  NVMK_REQUIRE(h != nullptr && h->magic == 0x5441424c && groups != nullptr, "nvmk_ff_tables_view: not a term-table handle");
  std::memset(groups, 0, sizeof(nvmk_ff_group) * 12);
  const Build& b = h->build;
  for (int g = 0; g < h->nGroups; ++g) view_group(b, b.groups[static_cast<size_t>(g)], &groups[g]);
  if (b.mergedGroup >= 0 && b.mergeImpossible.load() == 0) view_group(b, b.groups[static_cast<size_t>(b.mergedGroup)], &groups[11]);
  if (n_mols != nullptr) *n_mols = b.nMols;
  return NVMK_OK;
}

int nvmk_ff_tables_wait(const void* handle, void* stream) {
  TablesHandle* h = static_cast<TablesHandle*>(const_cast<void*>(handle));
  NVMK_REQUIRE(h != nullptr && h->magic == 0x5441424c, "nvmk_ff_tables_wait: not a term-table handle");
  return build_wait(h->build, -1, nvmk::as_stream(stream), "nvmk_ff_tables_wait");
}

int nvmk_ff_tables_free(void* handle) {
  if (handle == nullptr) return NVMK_OK;
  TablesHandle* h = static_cast<TablesHandle*>(handle);
  NVMK_REQUIRE(h->magic == 0x5441424c, "nvmk_ff_tables_free: not a term-table handle");
  h->magic = 0;
  if (h->build.builder.joinable()) h->build.builder.join();
  delete h;
  return NVMK_OK;
}

}  // extern "C"
