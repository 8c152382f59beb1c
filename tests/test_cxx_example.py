"""examples/cross_tanimoto_from_cxx.cpp — a C++ caller of the C ABI of the kind INTEGRATION.md describes: it must build against
the header and the library (CPU), and on a GPU reproduce a host popcount loop bit for bit."""

import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def build(tmp_path):
    if shutil.which("g++") is None or not Path("/opt/rocm/include/hip/hip_runtime.h").exists():
        pytest.skip("needs g++ and the HIP headers")
    exe = tmp_path / "cross_tanimoto"
    lib_dir = ROOT / "nvmolkit_amd" / "lib"
    run = subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", f"-I{ROOT / 'include'}", "-I/opt/rocm/include",
                          str(ROOT / "examples" / "cross_tanimoto_from_cxx.cpp"), f"-L{lib_dir}", "-lnvmolkit_amd", "-L/opt/rocm/lib",
                          "-lamdhip64", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)],
                         capture_output=True, text=True)
    assert run.returncode == 0, run.stderr[-2000:]
    return exe


def test_the_example_builds_against_header_and_library(tmp_path):
    build(tmp_path)


@pytest.mark.gpu
def test_the_example_reproduces_the_host_loop(tmp_path):
    run = subprocess.run([str(build(tmp_path))], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "0 differ from the host loop" in run.stdout
