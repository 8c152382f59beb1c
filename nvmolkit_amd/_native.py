"""ctypes binding of libnvmolkit_amd.so (the C ABI declared in include/nvmolkit_amd.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load, every
operator raises.  ``import torch`` happens before the library is opened so that the HIP runtime
torch ships (same SONAME, libamdhip64.so.7) is the one both share — device pointers and streams
created by torch are then directly usable by the kernels.
"""

from __future__ import annotations

import contextlib
import ctypes
import os
from pathlib import Path

import torch  # noqa: F401  (must be imported before the HIP library is dlopen'ed, see module docstring)

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "lib" / "libnvmolkit_amd.so"

OK = 0
TRUNCATED = 1  # success with a notice (nvmk_smiles_self_matches: the search gave up before the list was complete)
ERR_INVALID_ARGUMENT = -1
ERR_HIP = -2
ERR_OUT_OF_MEMORY = -3
ERR_UNSUPPORTED = -4
ERR_INTERNAL = -5

METRIC_TANIMOTO = 0
METRIC_COSINE = 1


class NativeLibraryError(ImportError):
    """libnvmolkit_amd.so is missing or could not be loaded."""


_lib: ctypes.CDLL | None = None

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int

# name -> (restype, argtypes); every symbol declared in include/nvmolkit_amd.h appears here.
SIGNATURES: dict[str, tuple] = {
    "nvmk_last_error": (ctypes.c_char_p, []),
    "nvmk_abi_version": (_int, []),
    "nvmk_device_count": (_int, [ctypes.POINTER(_int)]),
    "nvmk_device_memory": (_int, [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "nvmk_set_devices": (_int, [_vp, _int]),
    "nvmk_get_devices": (_int, [_vp, _int, ctypes.POINTER(_int)]),
    "nvmk_copy_peer_async": (_int, [_vp, _int, _vp, _vp, _int, _vp, ctypes.c_size_t]),
    "nvmk_comm_unique_id": (_int, [_vp]),
    "nvmk_comm_init_rank": (_int, [_vp, _int, _vp, _int]),
    "nvmk_comm_destroy": (_int, [_vp]),
    "nvmk_allgather_rows": (_int, [_vp, _vp, _i64, _int, _vp, _vp]),
    "nvmk_set_option": (_int, [ctypes.c_char_p, ctypes.c_char_p]),
    "nvmk_get_option": (_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]),
    "nvmk_cross_tanimoto_f64": (_int, [_vp, _i64, _vp, _i64, _int, _vp, _i64, _vp]),
    "nvmk_cross_cosine_f64": (_int, [_vp, _i64, _vp, _i64, _int, _vp, _i64, _vp]),
    "nvmk_fp4_workspace_bytes": (ctypes.c_size_t, [_i64, _int]),
    "nvmk_fp4_prepare": (_int, [_vp, _i64, _int, _vp, _vp]),
    "nvmk_cross_similarity_prepared_f64": (_int, [_int, _vp, _i64, _i64, _i64, _vp, _i64, _int, _vp, _i64, _vp]),
    "nvmk_cross_similarity_host_f64": (_int, [_int, _vp, _i64, _vp, _i64, _int, _vp, _i64]),
    "nvmk_neighbor_counts": (_int, [_int, _vp, _vp, _i64, _vp, _vp, _i64, _int, ctypes.c_float, _int, _vp, _vp]),
    "nvmk_butina_fused": (_int, [_int, _vp, _i64, _int, ctypes.c_double, _vp, _vp, _vp, ctypes.POINTER(_i64), _vp]),
    "nvmk_butina_pairs": (_int, [_int, _vp, _i64, _int, ctypes.c_double, _int, _int, _vp, _vp, ctypes.c_uint64,
                                 ctypes.POINTER(ctypes.c_uint64), _vp]),
    "nvmk_butina_from_pairs": (_int, [_i64, _vp, _vp, ctypes.c_uint64, _vp, _vp, _vp, ctypes.POINTER(_i64), _vp]),
    "nvmk_morgan_from_invariants": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _vp, _vp]),
    "nvmk_ff_energy": (_int, [_vp, ctypes.c_double, ctypes.c_double, _vp, _vp, _vp, _vp]),
    "nvmk_ff_gradient": (_int, [_vp, ctypes.c_double, ctypes.c_double, _vp, _vp, _vp, _vp]),
    "nvmk_bfgs_minimize": (_int, [_vp, _vp, ctypes.c_double, ctypes.c_double, _int, ctypes.c_double, _int, _vp, _vp, _vp,
                                  _vp, _vp, _vp]),
    "nvmk_bfgs_minimize_repeat": (_int, [_vp, _vp, ctypes.c_double, ctypes.c_double, _int, _int, ctypes.c_double, _int, _vp, _vp,
                                         _vp, _vp, _vp, _vp]),
    "nvmk_bfgs_minimize_two_stages": (_int, [_vp, _vp, ctypes.c_double, ctypes.c_double, _int, _int, _vp, ctypes.c_double, _int, _vp,
                                             _vp, _vp, _vp, _vp, _vp]),
    "nvmk_bfgs_set_stats": (_int, [_vp]),
    "nvmk_scheduler_create": (_vp, [_int, _int, _int]),
    "nvmk_scheduler_destroy": (None, [_vp]),
    "nvmk_scheduler_dispatch": (_int, [_vp, _int, _vp, ctypes.POINTER(_int)]),
    "nvmk_scheduler_record": (_int, [_vp, _vp, _vp, _int]),
    "nvmk_etkdg_embed": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "nvmk_etkdg_molset_build": (_int, [_vp, ctypes.c_int32, _int, ctypes.c_uint, _vp, ctypes.POINTER(ctypes.c_void_p)]),
    "nvmk_etkdg_molset_view": (_int, [_vp, _vp]),
    "nvmk_etkdg_molset_wait": (_int, [_vp, ctypes.c_int32, _vp]),
    "nvmk_ff_tables_wait": (_int, [_vp, _vp]),
    "nvmk_etkdg_molset_free": (_int, [_vp]),
    "nvmk_ff_tables_build": (_int, [_int, _vp, ctypes.c_int32, _int, _int, ctypes.c_uint, _vp, ctypes.POINTER(ctypes.c_void_p)]),
    "nvmk_ff_tables_view": (_int, [_vp, _vp, ctypes.POINTER(ctypes.c_int32)]),
    "nvmk_ff_tables_free": (_int, [_vp]),
    "nvmk_etkdg_stage_timings": (_int, [_vp, _vp, _vp, _vp, _int, _vp]),
    "nvmk_etkdg_random_coords": (_int, [ctypes.c_uint64, ctypes.c_uint64, _int, _vp, _vp, ctypes.c_double, _vp, _vp]),
    "nvmk_etkdg_driver_run": (_int, [_int, _int, _int, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_int32),
                                     ctypes.POINTER(ctypes.c_int32), _vp]),
    "nvmk_etkdg_stereo_check": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "nvmk_conformer_rmsd_batch": (_int, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int64, ctypes.c_int, _vp, _vp]),
    "nvmk_conformer_rmsd_batch_sym": (_int, [_vp, _vp, _vp, _vp, _int, _i64, _vp, _vp, _vp, _vp, _vp]),
    "nvmk_conformer_prune": (_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_double, _vp, _vp]),
    "nvmk_butina_dense": (_int, [_vp, _vp, _i64, ctypes.c_double, _int, _vp, _vp, ctypes.POINTER(_i64), _vp]),
    "nvmk_smiles_parse": (_int, [ctypes.POINTER(ctypes.c_char_p), _i64, _int, ctypes.POINTER(ctypes.c_void_p)]),
    "nvmk_smiles_parse_flags": (_int, [ctypes.POINTER(ctypes.c_char_p), _i64, _int, ctypes.c_uint, ctypes.POINTER(ctypes.c_void_p)]),
    "nvmk_smiles_parse_text": (_int, [ctypes.c_char_p, _i64, _int, ctypes.c_uint, ctypes.POINTER(ctypes.c_void_p)]),
    "nvmk_sdf_parse_text": (_int, [ctypes.c_char_p, _i64, _int, ctypes.c_uint, ctypes.POINTER(ctypes.c_void_p)]),
    "nvmk_smiles_size": (_int, [_vp, ctypes.POINTER(_i64)]),
    "nvmk_smiles_free": (_int, [_vp]),
    "nvmk_smiles_counts": (_int, [_vp, _vp, _vp, _vp]),
    "nvmk_smiles_self_matches": (_int, [_vp, _i64, _int, _int, _vp, _vp]),
    "nvmk_smiles_graph": (_int, [_vp, _i64, _vp, _vp]),
    "nvmk_smiles_morgan_inputs": (_int, [_vp, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _int]),
}


class FFGroup(ctypes.Structure):
    _fields_ = [("starts", ctypes.c_void_p), ("idx", ctypes.c_void_p), ("par", ctypes.c_void_p)]


class FFBatch(ctypes.Structure):
    """Mirror of ``nvmk_ff_batch`` (include/nvmolkit_amd.h)."""

    _fields_ = [("kind", ctypes.c_int32), ("n_systems", ctypes.c_int32), ("atom_starts", ctypes.c_void_p),
                ("groups", FFGroup * 12), ("system_mol", ctypes.c_void_p), ("group_mask", ctypes.c_uint32),
                ("etk_ref12_starts", ctypes.c_void_p), ("etk_ref12", ctypes.c_void_p),
                ("etk_ref13_starts", ctypes.c_void_p), ("etk_ref13", ctypes.c_void_p)]


class BfgsSecondStage(ctypes.Structure):
    """Mirror of ``nvmk_bfgs_second_stage``."""

    _fields_ = [("w0", ctypes.c_double), ("w1", ctypes.c_double), ("max_iters", ctypes.c_int32), ("restarts", ctypes.c_int32),
                ("d_pos_between", ctypes.c_void_p), ("skip_above_energy_per_atom", ctypes.c_double)]


FF_DG, FF_ETK, FF_MMFF, FF_QUARTIC, FF_UFF = 0, 1, 2, 3, 4


class EtkdgMolset(ctypes.Structure):
    """Mirror of ``nvmk_etkdg_molset``."""

    _fields_ = [("n_mols", ctypes.c_int32), ("h_n_atoms", ctypes.c_void_p), ("dg", FFGroup * 3), ("etk", FFGroup * 6),
                ("check_starts", ctypes.c_void_p), ("check_kind", ctypes.c_void_p), ("check_idx", ctypes.c_void_p),
                ("check_par", ctypes.c_void_p), ("num_impropers", ctypes.c_void_p),
                ("h_etk_d12_counts", ctypes.c_void_p), ("h_etk_d13_counts", ctypes.c_void_p), ("build_handle", ctypes.c_void_p)]


class EtkdgParams(ctypes.Structure):
    """Mirror of ``nvmk_etkdg_params``."""

    _fields_ = [("confs_per_mol", ctypes.c_int32), ("max_iterations", ctypes.c_int32), ("batch_size", ctypes.c_int32),
                ("use_exp_torsions", ctypes.c_int32), ("use_basic_knowledge", ctypes.c_int32),
                ("enforce_chirality", ctypes.c_int32), ("box_size", ctypes.c_double), ("force_tol", ctypes.c_double),
                ("seed", ctypes.c_uint64), ("batches_per_gpu", ctypes.c_int32)]


class HostTerms(ctypes.Structure):
    """Mirror of ``nvmk_host_terms``: one molecule's rows of one term group, in the caller's own arrays."""

    _fields_ = [("n_terms", ctypes.c_int32), ("idx_bytes", ctypes.c_int32), ("idx", ctypes.c_void_p), ("par", ctypes.c_void_p)]


class FlatMoleculeDesc(ctypes.Structure):
    """Mirror of ``nvmk_flat_molecule``."""

    _fields_ = [("n_atoms", ctypes.c_int32), ("num_impropers", ctypes.c_int32), ("has_etk", ctypes.c_int32), ("n_checks", ctypes.c_int32),
                ("dg", HostTerms * 3), ("etk", HostTerms * 6), ("check_kind", ctypes.c_void_p), ("check_idx", ctypes.c_void_p),
                ("check_par", ctypes.c_void_p)]


BUILD_KEEP_PAIR_ORDER, BUILD_NO_MMFF_MERGE, BUILD_HOST, BUILD_ASYNC = 1, 2, 4, 8

CHECK_TETRAHEDRAL, CHECK_CHIRAL_VOLUME, CHECK_CHIRAL_DISTANCE = 0, 1, 2
CHECK_CHIRAL_CENTER_VOLUME, CHECK_DOUBLE_BOND_STEREO, CHECK_DOUBLE_BOND_GEOMETRY = 3, 4, 5
ETKDG_N_STAGES = 11


def lib() -> ctypes.CDLL:
    """Load (once) and return the native library; raises NativeLibraryError if it is not built."""
    global _lib
    if _lib is None:
        path = Path(os.environ.get("NVMOLKIT_AMD_LIB", LIB_PATH))
        if not path.exists():
            raise NativeLibraryError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `python nvmolkit_amd/_build.py`); there is no CPU fallback."
            )
        try:
            L = ctypes.CDLL(str(path))
        except OSError as exc:  # pragma: no cover - depends on the host
            raise NativeLibraryError(f"could not load {path}: {exc}") from exc
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


PYGLUE_PATH = _PKG / "lib" / "_nvmk_pyglue.so"
_pyglue: ctypes.PyDLL | None = None


def pyglue() -> ctypes.PyDLL:
    """The CPython glue (pyglue/gather.c): fills ``nvmk_flat_molecule`` / ``nvmk_host_terms`` arrays from lists of Python molecule
    descriptions without a Python-level loop.  Loaded with ``PyDLL`` — its functions run with the GIL held and raise Python
    exceptions themselves.  Built next to the product library; missing = not built, there is no slower stand-in."""
    global _pyglue
    if _pyglue is None:
        if not PYGLUE_PATH.exists():
            raise NativeLibraryError(f"{PYGLUE_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        G = ctypes.PyDLL(str(PYGLUE_PATH))
        G.nvmk_py_gather_flat_molecules.restype = ctypes.c_int64
        G.nvmk_py_gather_flat_molecules.argtypes = [ctypes.py_object, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.py_object, ctypes.py_object]
        G.nvmk_py_gather_term_tables.restype = _int
        G.nvmk_py_gather_term_tables.argtypes = [ctypes.py_object, _vp, _vp, _int, _vp, ctypes.py_object, ctypes.py_object]
        _pyglue = G
    return _pyglue


def _as_term_array(obj, is_par: int):
    """Slow path of the glue: anything that is not already a C-contiguous int32 / int64 / float64 array."""
    import numpy as np

    return np.ascontiguousarray(obj, dtype=np.float64 if is_par else np.int32)


def build_flags() -> int:
    """``NVMK_PAIR_ORDER=input`` / ``NVMK_MMFF_MERGE=0`` (A/B switches of DESIGN.md section 5) as nvmk_*_build flags."""
    flags = 0
    if os.environ.get("NVMK_PAIR_ORDER", "diagonal") == "input":
        flags |= BUILD_KEEP_PAIR_ORDER
    if os.environ.get("NVMK_MMFF_MERGE", "1") == "0":
        flags |= BUILD_NO_MMFF_MERGE
    return flags


def build_threads(requested: int) -> int:
    """Host threads for the table builder: the caller's number if positive, else this process's share of the cores it may run
    on — one process per GPU means LOCAL_WORLD_SIZE processes assemble tables on one host at the same time (torchrun sets it), and
    eight of them asking for 64 threads each would oversubscribe a 128-core box inside everybody's timed region.  At most 64
    (the library's own limit)."""
    if requested > 0:
        return int(requested)
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    try:
        local = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        local = 1
    return max(1, min(64, cores // local))


def last_error() -> str:
    msg = lib().nvmk_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str = "") -> None:
    """Translate an ABI return code into the exception type the reference's bindings raise."""
    if rc == OK:
        return
    msg = last_error() or f"{what} failed with code {rc}"
    if rc == ERR_INVALID_ARGUMENT:
        raise ValueError(msg)  # reference: std::invalid_argument -> ValueError (Boost.Python translation)
    if rc == ERR_OUT_OF_MEMORY:
        raise RuntimeError(msg)  # reference: std::runtime_error("Not enough memory ...")
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)


def set_option(name: str, value) -> None:
    """Set a library switch (``NVMK_*`` of DESIGN.md); ``None`` unsets it.  The environment is only read once per process."""
    check(lib().nvmk_set_option(name.encode(), None if value is None else str(value).encode()), "nvmk_set_option")


def get_option(name: str) -> str:
    buf = ctypes.create_string_buffer(64)
    check(lib().nvmk_get_option(name.encode(), buf, len(buf)), "nvmk_get_option")
    return buf.value.decode()


@contextlib.contextmanager
def options(**values):
    """``with options(NVMK_SIM_PATH="valu"): ...`` — set switches for a block, then restore what they held."""
    old = {k: get_option(k) for k in values}
    try:
        for k, v in values.items():
            set_option(k, v)
        yield
    finally:
        for k, v in old.items():
            set_option(k, v if v != "" else None)


@contextlib.contextmanager
def on_stream(stream, device):
    """Run the staging of a call (allocations, ``.contiguous()`` / host-to-device copies) on the SAME stream as its kernel.

    With ``stream=None`` everything is on torch's current stream already.  With a side stream, staging done on the current
    stream would race with the kernel (no dependency between the streams) and its temporaries would return to the caching
    allocator tied to the wrong stream while the kernel may still read them: so the side stream first waits for the work
    already queued on the current stream (the inputs), then becomes current for the block."""
    if stream is None:
        yield
        return
    with torch.cuda.device(device):
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            yield


def stream_ptr(stream) -> int:
    """``torch.cuda.Stream | None`` -> raw hipStream_t value (reference: nvmolkit/similarity.py:66)."""
    if stream is not None and not isinstance(stream, torch.cuda.Stream):
        raise TypeError(f"stream must be a torch.cuda.Stream or None, got {type(stream).__name__}")
    s = stream if stream is not None else torch.cuda.current_stream()
    return int(s.cuda_stream)
