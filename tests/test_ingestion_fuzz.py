"""The SMILES / SDF ingestion reads untrusted text on the host: mutated real inputs through every entry point of the path under
AddressSanitizer + UndefinedBehaviorSanitizer (tests/native/fuzz_ingestion.cpp).  No GPU: smiles.cpp and runtime.cpp are
compiled for the host with g++."""

import os
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HIP_INCLUDE = Path("/opt/rocm/include")


@pytest.fixture(scope="module")
def fuzz_binary(tmp_path_factory):
    if shutil.which("g++") is None or not HIP_INCLUDE.exists():
        pytest.skip("needs g++ and the HIP headers")
    out = tmp_path_factory.mktemp("fuzz") / "fuzz_ingestion"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__",
           f"-I{ROOT / 'include'}", f"-I{HIP_INCLUDE}", str(ROOT / "tests" / "native" / "fuzz_ingestion.cpp"),
           str(ROOT / "nvmolkit_amd" / "csrc" / "smiles.cpp"), str(ROOT / "nvmolkit_amd" / "csrc" / "runtime.cpp"), "-o", str(out),
           "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"]
    built = subprocess.run(cmd, capture_output=True, text=True)
    if built.returncode != 0:
        pytest.skip("sanitizer build not available here: " + built.stderr[-300:])
    return out


@pytest.mark.parametrize("seed", [1, 2])
def test_mutated_smiles_and_sdf_run_clean_under_sanitizers(fuzz_binary, seed):
    golden = ROOT / "tests" / "golden"
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=0")
    run = subprocess.run([str(fuzz_binary), str(golden / "chembl_10k.smi"), "60", str(seed), str(golden / "MMFF94_dative_first5.sdf"),
                          str(golden / "larger_molecules.sdf")], capture_output=True, text=True, env=env, timeout=600)
    report = run.stdout + run.stderr
    assert run.returncode == 0, report[-2000:]
    assert "runtime error" not in report and "AddressSanitizer" not in report and "LeakSanitizer" not in report, report[-2000:]
    assert "done 15360 mutated SMILES" in report
