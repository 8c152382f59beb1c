"""CPU oracle for the nvMolKit hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``nvmolkit_amd``) never does.  The arithmetic lives in the C files next
to this module (each function cites the reference lines it restates); this module is the ctypes
loader plus numpy-friendly wrappers.
"""

from __future__ import annotations

import ctypes
import hashlib
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB_PATH = _DIR / "liboracle.so"

TANIMOTO = 0
COSINE = 1


def _host_stamp() -> str:
    """The library is built with -march=native, so it is rebuilt when the host CPU changes
    (the build container and the GPU box have different CPUs)."""
    try:
        txt = Path("/proc/cpuinfo").read_text()
        keep = [ln for ln in txt.splitlines() if ln.startswith(("model name", "flags"))][:2]
        return hashlib.sha256("\n".join(keep).encode()).hexdigest()
    except OSError:
        return "unknown"


def build(force: bool = False) -> Path:
    """Compile oracle/*.c with gcc (a few seconds)."""
    srcs = sorted(_DIR.glob("oracle_*.c"))
    stamp = _DIR / "liboracle.stamp"
    want = _host_stamp()
    stale = (force or not _LIB_PATH.exists() or not stamp.exists() or stamp.read_text() != want or
             any(s.stat().st_mtime > _LIB_PATH.stat().st_mtime for s in srcs))
    if stale:
        subprocess.run(["make", "-C", str(_DIR), "-B", "liboracle.so"], check=True, capture_output=True)
        stamp.write_text(want)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(_LIB_PATH))
        _declare(_lib)
    return _lib


_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def _declare(L: ctypes.CDLL) -> None:
    L.orc_num_threads.restype = ctypes.c_int
    L.orc_set_num_threads.argtypes = [ctypes.c_int]
    L.orc_set_num_threads.restype = None
    L.orc_cross_similarity_f64.argtypes = [ctypes.c_int, _u32p, ctypes.c_int64, _u32p, ctypes.c_int64, ctypes.c_int,
                                           _f64p, ctypes.c_int64, ctypes.c_int]
    L.orc_cross_similarity_f64.restype = None
    L.orc_have_vpopcnt.restype = ctypes.c_int
    L.orc_set_scalar.argtypes = [ctypes.c_int]
    L.orc_set_scalar.restype = None
    L.orc_cross_intersection_i32.argtypes = [_u32p, ctypes.c_int64, _u32p, ctypes.c_int64, ctypes.c_int, _i32p]
    L.orc_cross_intersection_i32.restype = None
    L.orc_neighbor_counts.argtypes = [ctypes.c_int, _u32p, ctypes.c_int64, _u32p, ctypes.c_int64, ctypes.c_int,
                                      ctypes.c_float, ctypes.c_int, _i32p]
    L.orc_neighbor_counts.restype = None
    L.orc_butina_fused.argtypes = [ctypes.c_int, _u32p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, _i32p, _i64p,
                                   _i32p]
    L.orc_butina_fused.restype = ctypes.c_int64
    L.orc_butina_from_pairs.argtypes = [ctypes.c_int64, _i32p, _i32p, ctypes.c_int64, _i32p, _i64p, _i32p]
    L.orc_butina_from_pairs.restype = ctypes.c_int64
    L.orc_butina_dense.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, _i32p,
                                   ctypes.c_void_p]
    L.orc_butina_dense.restype = ctypes.c_int64
    _i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
    L.orc_morgan_hash_vector.argtypes = [_u32p, ctypes.c_int]
    L.orc_morgan_hash_vector.restype = ctypes.c_uint32
    L.orc_morgan_environments.argtypes = [_u32p, _u32p, _i16p, _i16p, ctypes.c_int, ctypes.c_int, _u32p, _i32p]
    L.orc_morgan_environments.restype = ctypes.c_int
    L.orc_morgan_fingerprints.argtypes = [_u32p, _u32p, _i16p, _i16p, _i16p, ctypes.c_int64, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, _u32p]
    L.orc_morgan_fingerprints.restype = None
    L.orc_check_newton_division.argtypes = [ctypes.c_int]
    L.orc_check_newton_division.restype = ctypes.c_int64
    L.orc_check_threshold_arith.argtypes = [ctypes.c_float, ctypes.c_int]
    L.orc_check_threshold_arith.restype = ctypes.c_int64


def _as_u32(x) -> np.ndarray:
    a = np.ascontiguousarray(x)
    if a.dtype == np.int32:
        a = a.view(np.uint32)
    if a.dtype != np.uint32:
        raise TypeError(f"fingerprints must be uint32/int32 words, got {a.dtype}")
    if a.ndim != 2:
        raise ValueError("fingerprints must be 2-D (n, words)")
    return a


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_num_threads(n: int) -> None:
    """Threads of the oracle's OpenMP regions from now on (<= 0: all processors)."""
    lib().orc_set_num_threads(int(n))


def cross_similarity(a, b=None, metric: int = TANIMOTO, threads: int = 0) -> np.ndarray:
    """N x M float64 similarity matrix (reference: src/similarity_kernels.cu:350-364)."""
    a = _as_u32(a)
    b = a if b is None else _as_u32(b)
    if a.shape[1] != b.shape[1]:
        raise ValueError("fingerprint width mismatch")
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float64)
    lib().orc_cross_similarity_f64(metric, a, a.shape[0], b, b.shape[0], a.shape[1], out, b.shape[0], threads)
    return out


def cross_intersection(a, b=None) -> np.ndarray:
    a = _as_u32(a)
    b = a if b is None else _as_u32(b)
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.int32)
    lib().orc_cross_intersection_i32(a, a.shape[0], b, b.shape[0], a.shape[1], out)
    return out


def neighbor_counts(x, y, threshold: float, sign: int = 1, metric: int = TANIMOTO, counts=None) -> np.ndarray:
    x = _as_u32(x)
    y = _as_u32(y)
    if counts is None:
        counts = np.zeros(x.shape[0], dtype=np.int32)
    lib().orc_neighbor_counts(metric, x, x.shape[0], y, y.shape[0], x.shape[1], np.float32(threshold), sign, counts)
    return counts


def check_newton_division(umax: int) -> int:
    """Number of (c, u) pairs where the rcp_f32 + Newton division shortcut differs from IEEE c / u (must be 0)."""
    return int(lib().orc_check_newton_division(int(umax)))


def check_threshold_arith(thr: float, smax: int) -> int:
    """Number of (s, c) pairs where the exact-arithmetic Tanimoto threshold predicate of the ring count kernel differs
    from the float-division predicate `float(c)/float(s-c) >= thr` (must be 0)."""
    return int(lib().orc_check_threshold_arith(float(thr), int(smax)))


def butina_fused(x, cutoff: float, metric: int = TANIMOTO):
    """Returns (clusters: list[tuple[int]], cumulative sizes, centroids) like fused_butina."""
    x = _as_u32(x)
    n = x.shape[0]
    idx = np.zeros(max(n, 1), dtype=np.int32)
    offs = np.zeros(n + 1, dtype=np.int64)
    cent = np.zeros(max(n, 1), dtype=np.int32)
    nc = lib().orc_butina_fused(metric, x, n, x.shape[1], float(cutoff), idx, offs, cent)
    clusters = [tuple(int(v) for v in idx[offs[k]:offs[k + 1]]) for k in range(nc)]
    return clusters, [int(v) for v in offs[:nc + 1]], [int(v) for v in cent[:nc]]


def butina_from_pairs(n: int, counts, pairs):
    """Fused-Butina rounds on a neighbour graph (degrees incl. self, unordered pairs once) -> like butina_fused."""
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    idx = np.zeros(max(n, 1), dtype=np.int32)
    offs = np.zeros(n + 1, dtype=np.int64)
    cent = np.zeros(max(n, 1), dtype=np.int32)
    nc = lib().orc_butina_from_pairs(n, counts, pairs, len(pairs), idx, offs, cent)
    clusters = [tuple(int(v) for v in idx[offs[k]:offs[k + 1]]) for k in range(nc)]
    return clusters, [int(v) for v in offs[:nc + 1]], [int(v) for v in cent[:nc]]


def neighbor_pairs(x, cutoff: float, row_lo: int = 0, row_hi: int | None = None, metric: int = TANIMOTO):
    """(partial counts int32[N], pairs int32[E, 2]) of the unordered neighbour pairs (i, j), i < j, whose FIRST row lies in
    [row_lo, row_hi), plus the self pairs of those rows — a row shard of the symmetric all-pairs pass; the predicate is the
    f32 one of the fused kernel (float(c) / float(u) >= float32(1 - cutoff))."""
    x = _as_u32(x)
    n = x.shape[0]
    row_hi = n if row_hi is None else row_hi
    thr = np.float32(1.0 - cutoff)
    inter = cross_intersection(x[row_lo:row_hi], x).astype(np.float32)
    pc = np.array([int(np.unpackbits(r.view(np.uint8)).sum()) for r in x], dtype=np.float32)
    if metric == TANIMOTO:
        union = pc[row_lo:row_hi, None] + pc[None, :] - inter
        with np.errstate(divide="ignore", invalid="ignore"):
            hit = np.where(union > 0, inter / np.maximum(union, np.float32(1)) >= thr, False)
    else:
        denom = np.sqrt(pc[row_lo:row_hi, None] * pc[None, :]).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            hit = np.where(denom > 0, inter / np.maximum(denom, np.float32(1e-30)) >= thr, False)
    counts = np.zeros(n, dtype=np.int32)
    ii, jj = np.nonzero(hit)
    ii = ii + row_lo
    keep = ii < jj
    pairs = np.stack([ii[keep], jj[keep]], 1).astype(np.int32)
    np.add.at(counts, pairs[:, 0], 1)
    np.add.at(counts, pairs[:, 1], 1)
    self_hit = ii == jj
    np.add.at(counts, ii[self_hit], 1)
    return counts, pairs


def butina_dense(dist=None, cutoff: float = 0.0, hit=None):
    """Returns (cluster ids int32[N], centroids int32[n_clusters])."""
    if dist is not None:
        d = np.ascontiguousarray(dist, dtype=np.float64)
        n = d.shape[0]
        dptr, hptr = d.ctypes.data, None
    else:
        h = np.ascontiguousarray(hit, dtype=np.uint8)
        n = h.shape[0]
        dptr, hptr = None, h.ctypes.data
    clusters = np.empty(max(n, 1), dtype=np.int32)
    cent = np.empty(max(n, 1), dtype=np.int32)
    nc = lib().orc_butina_dense(dptr, hptr, n, float(cutoff), clusters, cent.ctypes.data)
    return clusters[:n], cent[:nc]


# ---- numpy restatements (independent of the C code; used to cross-check it) -------------------


def unpack_bits(words) -> np.ndarray:
    """(n, W) u32 -> (n, 32*W) bool; bit j of a fingerprint = bit j%32 of word j//32
    (nvmolkit/fingerprints.py:25-72)."""
    w = _as_u32(words)
    shifts = np.arange(32, dtype=np.uint32)
    return ((w[:, :, None] >> shifts) & 1).astype(bool).reshape(w.shape[0], -1)


def pack_bits(bits) -> np.ndarray:
    b = np.ascontiguousarray(bits).astype(np.uint32)
    n, nb = b.shape
    pad = (-nb) % 32
    if pad:
        b = np.concatenate([b, np.zeros((n, pad), dtype=np.uint32)], axis=1)
    b = b.reshape(n, -1, 32)
    return (b << np.arange(32, dtype=np.uint32)).sum(axis=2, dtype=np.uint32)


def cross_similarity_numpy(a, b=None, metric: int = TANIMOTO) -> np.ndarray:
    """Bit-unpack restatement, as the reference's own test does (nvmolkit/tests/test_clustering.py:166-180)."""
    ua = unpack_bits(a).astype(np.int64)
    ub = ua if b is None else unpack_bits(b).astype(np.int64)
    inter = ua @ ub.T
    pa = ua.sum(1)[:, None]
    pb = ub.sum(1)[None, :]
    if metric == TANIMOTO:
        union = np.maximum(pa + pb - inter, 1)
        return inter.astype(np.float64) / union.astype(np.float64)
    denom = np.sqrt(pa.astype(np.float64) * pb.astype(np.float64))
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.where((inter == 0) | (denom == 0), 0.0, inter / denom)
    return out


# ---- Morgan fingerprints on flattened graphs (oracle_morgan.c) ---------------------------------


def morgan_hash_vector(components) -> int:
    """gboost::hash<std::vector<uint32_t>> with a 32-bit seed (src/morgan_fingerprint_common.cpp:54,121)."""
    v = np.ascontiguousarray(np.asarray(components, dtype=np.int64).astype(np.uint32))
    return int(lib().orc_morgan_hash_vector(v, len(v)))


def morgan_environments(atom_inv, bond_inv, bond_idx, bond_other, n_atoms: int, radius: int):
    """(codes, layers) of one molecule: the unfolded bit ids RDKit calls atom environments."""
    atom_inv = np.ascontiguousarray(atom_inv, dtype=np.uint32)
    bond_inv = np.ascontiguousarray(bond_inv, dtype=np.uint32)
    bond_idx = np.ascontiguousarray(bond_idx, dtype=np.int16)
    bond_other = np.ascontiguousarray(bond_other, dtype=np.int16)
    cap = max(1, (radius + 1) * max(n_atoms, 1))
    codes = np.zeros(cap, dtype=np.uint32)
    layers = np.zeros(cap, dtype=np.int32)
    n = lib().orc_morgan_environments(atom_inv, bond_inv, bond_idx, bond_other, n_atoms, radius, codes, layers)
    return codes[:n].copy(), layers[:n].copy()


def morgan_fingerprints(atom_inv, bond_inv, bond_idx, bond_other, n_atoms, stride: int, radius: int,
                        fp_bits: int) -> np.ndarray:
    """Batch in the ComputeInvariantsInto layout -> (n_mols, fp_bits/32) uint32 bit vectors."""
    n_atoms = np.ascontiguousarray(n_atoms, dtype=np.int16)
    n_mols = len(n_atoms)
    out = np.zeros((n_mols, fp_bits // 32), dtype=np.uint32)
    lib().orc_morgan_fingerprints(np.ascontiguousarray(atom_inv, dtype=np.uint32),
                                  np.ascontiguousarray(bond_inv, dtype=np.uint32),
                                  np.ascontiguousarray(bond_idx, dtype=np.int16),
                                  np.ascontiguousarray(bond_other, dtype=np.int16), n_atoms, n_mols, stride, radius,
                                  fp_bits, out)
    return out
