"""The C-ABI library builds, loads without a GPU and exports every symbol include/*.h declares."""

import ctypes
import re
from pathlib import Path

from nvmolkit_amd import _native

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "nvmolkit_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nvmk_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    names = declared_symbols()
    assert "nvmk_cross_tanimoto_f64" in names and len(names) >= 8


def test_every_declared_symbol_is_exported(native_lib):
    for name in declared_symbols():
        assert hasattr(native_lib, name), f"{name} declared in include/nvmolkit_amd.h but not exported"


def test_python_binding_covers_header(native_lib):
    assert sorted(_native.SIGNATURES) == declared_symbols()


def test_abi_version_and_error_slot(native_lib):
    assert native_lib.nvmk_abi_version() >= 1
    # argument validation happens before any HIP call, so it works without a GPU
    rc = native_lib.nvmk_cross_tanimoto_f64(None, 4, None, 4, 100, None, 4, None)
    assert rc == _native.ERR_INVALID_ARGUMENT
    assert b"multiple of 32" in native_lib.nvmk_last_error()
    n = ctypes.c_int64(-1)
    rc = native_lib.nvmk_butina_dense(None, None, 0, 0.1, 7, None, None, ctypes.byref(n), None)
    assert rc == _native.ERR_INVALID_ARGUMENT and b"neighborlistMaxSize" in native_lib.nvmk_last_error()
    # zero-size problems are no-ops
    assert native_lib.nvmk_cross_cosine_f64(None, 0, None, 5, 2048, None, 5, None) == _native.OK


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import pytest

    monkeypatch.setenv("NVMOLKIT_AMD_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_native, "_lib", None)
    with pytest.raises(_native.NativeLibraryError):
        _native.lib()


def test_null_and_out_of_range_arguments_never_crash():
    """Every entry point called with NULL pointers and 0 / 1 / -1 / 64 in every integer slot, in a child process: an error
    code (or a no-op) is fine, a signal is not.  Works without a GPU: arguments are checked before anything touches HIP."""
    import subprocess
    import sys

    code = r'''
import ctypes, sys
from nvmolkit_amd import _native
lib = _native.lib()
ints = (ctypes.c_int, ctypes.c_int64, ctypes.c_uint, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_int32)
calls = 0
for name, (restype, argtypes) in sorted(_native.SIGNATURES.items()):
    for variant in range(4):
        args = [[0, 1, -1, 64][variant] if t in ints else [0.0, 0.5, -1.0, 1e300][variant] if t in (ctypes.c_double, ctypes.c_float) else None
                for t in argtypes]
        getattr(lib, name)(*args)
        calls += 1
print("calls", calls)
'''
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=str(ROOT))
    assert p.returncode == 0, (p.returncode, p.stderr[-2000:])
    assert int(p.stdout.split()[-1]) == 4 * len(_native.SIGNATURES)


def test_integration_notes_name_every_entry_point():
    """INTEGRATION.md shows how each entry of the header would be bound from the reference's side: none may be missing."""
    import re

    header = (ROOT / "include" / "nvmolkit_amd.h").read_text()
    notes = (ROOT / "INTEGRATION.md").read_text()
    symbols = sorted(set(re.findall(r"\b(nvmk_[a-z0-9_]+)\s*\(", header)))
    assert len(symbols) >= 40
    assert [s for s in symbols if s not in notes] == []


def test_header_is_plain_c_and_cxx(tmp_path):
    """The boundary is a C ABI: the header compiles as C99 (pedantic) and as C++11, and a C program that takes the address of
    every declared entry links against the library."""
    import re
    import shutil
    import subprocess

    import pytest

    if shutil.which("gcc") is None or shutil.which("g++") is None:
        pytest.skip("needs gcc and g++")
    header = ROOT / "include" / "nvmolkit_amd.h"
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", str(header)],
                ["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", str(header)]):
        run = subprocess.run(cmd, capture_output=True, text=True)
        assert run.returncode == 0, run.stderr[-1500:]
    symbols = sorted(set(re.findall(r"\b(nvmk_[a-z0-9_]+)\s*\(", header.read_text())))
    src = tmp_path / "link_all.c"
    src.write_text('#include "nvmolkit_amd.h"\n#include <stdio.h>\nint main(void) {\n  const void* entries[] = {\n'
                   + "".join(f"    (const void*)&{s},\n" for s in symbols)
                   + '  };\n  printf("%d\\n", (int)(sizeof entries / sizeof entries[0]));\n  return 0;\n}\n')
    exe = tmp_path / "link_all"
    lib_dir = ROOT / "nvmolkit_amd" / "lib"
    run = subprocess.run(["gcc", "-std=c99", f"-I{ROOT / 'include'}", str(src), "-o", str(exe), f"-L{lib_dir}", "-lnvmolkit_amd",
                          f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr[-1500:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and int(out.stdout.strip()) == len(symbols), out.stdout + out.stderr
