"""The row-panel count kernel (csrc/count_panel.inc) depends on what the compiler does NOT do: its LDS traffic and its waits are
written by hand because the compiler, for any LDS access it can see, first waits for EVERY outstanding global -> LDS DMA
(`s_waitcnt vmcnt(0)`), which serialises the sixteen-buffer ring.  Round 5 met that twice: a build of an experiment compiled with
fourteen such waits inside the sweep and ran 24 % slower with the same arithmetic (profiles/r05_similarity/ab_half_k_prune_rejected.txt).
hipcc cross-compiles without a GPU, so the shipped source is held to its own numbers here: no scratch memory, the register budget of
one wave per SIMD, 16 matrix instructions per chunk step, and no more full DMA drains than the kernel's set-up and tear-down need."""

import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
SRC = ROOT / "nvmolkit_amd" / "csrc" / "similarity_mfma.hip"


@pytest.fixture(scope="module")
def assembly(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("panel_asm") / "similarity_mfma.s"
    run = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}", "-x", "hip", str(SRC),
                          "--cuda-device-only", "-S", "-o", str(out)], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    return out.read_text()


def kernel_body(asm: str, mangled_piece: str) -> str:
    start = re.search(rf"^(_ZN4nvmk3fp4\S*neighbor_count_panel_kernel{mangled_piece}\S*):", asm, re.M)
    assert start, mangled_piece
    end = asm.index(".amdhsa_kernel " + start.group(1), start.end())
    return asm[start.end():end]


def setting(asm: str, mangled_piece: str, name: str) -> int:
    m = re.search(rf"\.set \S*neighbor_count_panel_kernel{mangled_piece}\S*\.{name}, (\d+)", asm)
    assert m, (mangled_piece, name)
    return int(m.group(1))


@pytest.mark.parametrize("emit", ["ILb1E", "ILb0E"])
@pytest.mark.parametrize("ksteps,chunks", [("Li32E", 8), ("Li16E", 4), ("Li8E", 2)])
def test_compiled_row_panel_kernel_keeps_its_shape(assembly, emit, ksteps, chunks):
    piece = emit + ksteps
    body = kernel_body(assembly, piece)
    assert setting(assembly, piece, "private_seg_size") == 0                       # nothing spilled
    assert setting(assembly, piece, "num_vgpr") <= 256 and setting(assembly, piece, "num_agpr") <= 256
    # one copy of the sweep: 16 matrix instructions per chunk step + the 16 of the last chunk behind the loop
    assert len(re.findall(r"^\s*v_mfma_scale_f32_32x32x64_f8f6f4", body, re.M)) == 16 * (chunks + 1)
    # the ring's waits are the hand-written vmcnt(k) with k > 0; a drain of every outstanding DMA belongs to the panel's start, the pair
    # flushes and the sweep's end (19 in the 2048-bit pair-emitting form at this compiler) — an experiment that serialised the ring had 33
    drains = len(re.findall(r"s_waitcnt[^\n]*vmcnt\(0\)", body))
    assert drains <= 22, drains
    partial = len(re.findall(r"s_waitcnt[^\n]*vmcnt\([1-9]\d*\)", body))
    assert partial >= chunks, partial                                              # every chunk step waits for ITS chunk only
