"""examples/*.cpp — C++ callers of the C ABI of the kind INTEGRATION.md describes: they must build against the header and the
library (CPU), and on a GPU reproduce a host popcount loop bit for bit.  cross_tanimoto_from_cxx.cpp: one entry point;
sharded_reference_from_cxx.cpp: the configs[4] flow (communicator, all-gather of the reference shard, similarity launch), which a
one-GPU box runs as one rank; conformers_from_cxx.cpp: host term arrays -> table builder -> ETKDG -> MMFF minimisation."""

import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def build(tmp_path, name="cross_tanimoto_from_cxx"):
    if shutil.which("g++") is None or not Path("/opt/rocm/include/hip/hip_runtime.h").exists():
        pytest.skip("needs g++ and the HIP headers")
    exe = tmp_path / name
    lib_dir = ROOT / "nvmolkit_amd" / "lib"
    run = subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", f"-I{ROOT / 'include'}", "-I/opt/rocm/include",
                          str(ROOT / "examples" / f"{name}.cpp"), f"-L{lib_dir}", "-lnvmolkit_amd", "-L/opt/rocm/lib",
                          "-lamdhip64", "-lpthread", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)],
                         capture_output=True, text=True)
    assert run.returncode == 0, run.stderr[-2000:]
    return exe


def test_the_example_builds_against_header_and_library(tmp_path):
    build(tmp_path)
    build(tmp_path, "sharded_reference_from_cxx")
    build(tmp_path, "conformers_from_cxx")


@pytest.mark.gpu
def test_the_example_reproduces_the_host_loop(tmp_path):
    run = subprocess.run([str(build(tmp_path))], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "0 differ from the host loop" in run.stdout


@pytest.mark.gpu
def test_the_sharded_reference_example_runs_as_one_rank(tmp_path):
    run = subprocess.run([str(build(tmp_path, "sharded_reference_from_cxx"))], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "rank 0 of 1" in run.stdout and " 0 differ from the host loop" in run.stdout


@pytest.mark.gpu
def test_the_conformer_example_goes_from_host_arrays_to_minimised_conformers(tmp_path):
    """examples/conformers_from_cxx.cpp: per-molecule host arrays -> nvmk_etkdg_molset_build -> nvmk_etkdg_embed, the MMFF tables
    assembled on a second host thread meanwhile (nvmk_ff_tables_build) -> nvmk_bfgs_minimize; no Python in the process."""
    run = subprocess.run([str(build(tmp_path, "conformers_from_cxx"))], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "conformers from C++: OK" in run.stdout and " 0 with a distance-violation energy" in run.stdout
    assert " 0 whose energy did not go down" in run.stdout
