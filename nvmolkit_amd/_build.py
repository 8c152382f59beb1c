"""Build libnvmolkit_amd.so in-tree with hipcc for gfx950.

The library is a plain C-ABI shared object (see include/nvmolkit_amd.h); it is built with an
explicit hipcc command rather than torch.utils.cpp_extension because nothing in it depends on
torch.  hipcc cross-compiles without a GPU, so this runs in the CPU-only build container too.
"""

from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC_DIR = PKG_DIR / "csrc"
LIB_DIR = PKG_DIR / "lib"
LIB_PATH = LIB_DIR / "libnvmolkit_amd.so"
OBJ_DIR = PKG_DIR / "build"
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH and /opt/rocm/bin/hipcc)")


def sources() -> list[Path]:
    return sorted(list(CSRC_DIR.glob("*.hip")) + list(CSRC_DIR.glob("*.cpp")))


def _digest(paths: list[Path]) -> str:
    h = hashlib.sha256()
    for p in paths:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def _deps() -> list[Path]:
    return sorted(list(CSRC_DIR.glob("*.h")) + list(CSRC_DIR.glob("*.hpp")) + list(CSRC_DIR.glob("*.inc")) + [PKG_DIR.parent / "include" / "nvmolkit_amd.h"])


def build(force: bool = False, verbose: bool = False, variant: str = "") -> Path:
    """Compile every source under csrc/ into lib/libnvmolkit_amd.so (incremental per file).

    ``variant`` (or $NVMK_BUILD_VARIANT): an A/B build beside the product — lib/libnvmolkit_amd_<variant>.so from its own object
    directory, normally with $NVMK_EXTRA_HIPCC_FLAGS; a process picks it up through $NVMOLKIT_AMD_LIB (tools/ab_conformers.sh)."""
    variant = variant or os.environ.get("NVMK_BUILD_VARIANT", "")
    LIB_PATH = LIB_DIR / (f"libnvmolkit_amd_{variant}.so" if variant else "libnvmolkit_amd.so")
    OBJ_DIR = PKG_DIR / (f"build_{variant}" if variant else "build")
    LIB_DIR.mkdir(exist_ok=True)
    OBJ_DIR.mkdir(exist_ok=True)
    hipcc = _hipcc()
    common = [
        hipcc,
        f"--offload-arch={ARCH}",
        "-O3",
        "-std=c++17",
        "-fPIC",
        
        "-Wall",
        "-Wno-unused-function",
        f"-I{PKG_DIR.parent / 'include'}",
    ] + os.environ.get("NVMK_EXTRA_HIPCC_FLAGS", "").split()  # experiments (e.g. -DNVMK_BFGS_THREADS=512); part of the stamp
    dep_digest = _digest(_deps()) + hashlib.sha256(os.environ.get("NVMK_EXTRA_HIPCC_FLAGS", "").encode()).hexdigest()
    objs: list[Path] = []
    relink = force or not LIB_PATH.exists()
    stale: list[tuple[Path, Path, Path, str]] = []
    for src in sources():
        obj = OBJ_DIR / (src.name + ".o")
        stamp = OBJ_DIR / (src.name + ".sha")
        digest = _digest([src]) + dep_digest
        if force or not obj.exists() or not stamp.exists() or stamp.read_text() != digest:
            stale.append((src, obj, stamp, digest))
        objs.append(obj)

    def compile_one(job: tuple[Path, Path, Path, str]) -> None:
        src, obj, stamp, digest = job
        cmd = common + (["-x", "hip"] if src.suffix == ".hip" else []) + ["-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        stamp.write_text(digest)

    if stale:
        # the translation units are independent: compile them side by side (minimize.hip alone takes minutes — four thread
        # counts x seven force-field kinds of the fused BFGS kernel); the longest first
        from concurrent.futures import ThreadPoolExecutor

        stale.sort(key=lambda job: -job[0].stat().st_size if job[0].name != "minimize.hip" else -(1 << 40))
        jobs = max(1, min(len(stale), int(os.environ.get("NVMK_BUILD_JOBS", "0")) or (os.cpu_count() or 1)))
        with ThreadPoolExecutor(max_workers=jobs) as pool:
            list(pool.map(compile_one, stale))
        relink = True
    if relink:
        cmd = [
            hipcc,
            f"--offload-arch={ARCH}",
            "-shared",
            "-fPIC",
            
            "-o",
            str(LIB_PATH),
            *map(str, objs),
            "-Wl,-rpath,/opt/rocm/lib",
            "-Wl,--no-undefined",
            "-lpthread",
            "-ldl",
        ]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    if not variant:
        # The glue needs a C compiler and this interpreter's headers; the similarity, Butina and fingerprint paths do not use it.
        # A host without them still gets the library: _native.pyglue() raises when the conformer path first asks for the glue.
        try:
            build_pyglue(force=force, verbose=verbose)
        except (RuntimeError, subprocess.CalledProcessError, OSError) as exc:
            import warnings

            warnings.warn(f"nvmolkit_amd: the CPython glue (pyglue/gather.c) was not built: {exc}; FlatMoleculeSet / MoleculeTermTables "
                          "from Python molecule lists will raise until it is", RuntimeWarning)
    return LIB_PATH


PYGLUE_SRC = PKG_DIR / "pyglue" / "gather.c"
PYGLUE_PATH = LIB_DIR / "_nvmk_pyglue.so"


def build_pyglue(force: bool = False, verbose: bool = False) -> Path:
    """The CPython glue of the host layer (pyglue/gather.c: fills the C ABI's descriptor arrays from Python molecule lists) ->
    lib/_nvmk_pyglue.so, with the C compiler against this interpreter's headers.  It links against nothing: the Python symbols
    come from the interpreter that loads it (ctypes.PyDLL)."""
    import sysconfig

    LIB_DIR.mkdir(exist_ok=True)
    stamp = OBJ_DIR / "pyglue.sha"
    OBJ_DIR.mkdir(exist_ok=True)
    digest = _digest([PYGLUE_SRC, PKG_DIR.parent / "include" / "nvmolkit_amd.h"]) + sysconfig.get_python_version()
    if force or not PYGLUE_PATH.exists() or not stamp.exists() or stamp.read_text() != digest:
        cc = os.environ.get("CC") or shutil.which("gcc") or shutil.which("cc")
        if not cc:
            raise RuntimeError("no C compiler found for nvmolkit_amd/pyglue/gather.c")
        numpy_flags: list[str] = []
        try:  # numpy's C API for the array reads (gather.c); without its headers the buffer protocol serves
            import numpy

            if (Path(numpy.get_include()) / "numpy" / "arrayobject.h").exists():
                numpy_flags = ["-DNVMK_GLUE_NUMPY", f"-I{numpy.get_include()}"]
        except Exception:  # noqa: BLE001
            pass
        cmd = [cc, "-O2", "-std=c11", "-fPIC", "-shared", "-Wall", f"-I{sysconfig.get_paths()['include']}",
               f"-I{PKG_DIR.parent / 'include'}", *numpy_flags, str(PYGLUE_SRC), "-o", str(PYGLUE_PATH)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        stamp.write_text(digest)
    return PYGLUE_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
