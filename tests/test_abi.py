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
