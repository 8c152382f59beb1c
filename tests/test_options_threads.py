"""include/nvmolkit_amd.h: nvmk_set_option is safe against concurrent callers (the switches used to be read with
getenv on every call).  Writers and readers of the option registry under ThreadSanitizer (tests/native/options_race.cpp);
host code only, no GPU."""

import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HIP_INCLUDE = Path("/opt/rocm/include")


def test_option_registry_has_no_data_race_and_no_torn_values(tmp_path):
    if shutil.which("g++") is None or not HIP_INCLUDE.exists():
        pytest.skip("needs g++ and the HIP headers")
    out = tmp_path / "options_race"
    built = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", f"-I{ROOT / 'include'}",
                            f"-I{HIP_INCLUDE}", str(ROOT / "tests" / "native" / "options_race.cpp"),
                            str(ROOT / "nvmolkit_amd" / "csrc" / "runtime.cpp"), "-o", str(out), "-L/opt/rocm/lib", "-lamdhip64",
                            "-lpthread", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    if built.returncode != 0:
        pytest.skip("ThreadSanitizer build not available here: " + built.stderr[-300:])
    run = subprocess.run([str(out)], capture_output=True, text=True, timeout=300)
    report = run.stdout + run.stderr
    assert run.returncode == 0, report[-2000:]
    assert "ThreadSanitizer" not in report, report[-2000:]
    assert "torn 0" in report
