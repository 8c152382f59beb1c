"""The inverse-Hessian update as an operation of its own — the reference tests `updateInverseHessianBFGSBatch`
(src/minimizer/bfgs_hessian.cu) against a textbook loop in tests/test_bfgs_hessian.cpp:27-596: one system of 88 atoms, one of 300
("large"), four systems {3, 2, 33 or 300, 14} of which the third is inactive, and systems whose dGrad . xi has the wrong sign (no
update, only the new direction), each in 3 and 4 dimensions, from the identity and from a random symmetric matrix, to 1e-5.

Here the update is fused into the minimisation kernel, so tests/native/hess_update_check.hip drives the product's own pass
(csrc/hess_pass.h) through one update and the test compares it with oracle/ff.py:inverse_hessian_update — same cases, same
random shapes (uniform(-1, 1), signs of xi aligned with or against dGrad), 1e-10 instead of 1e-5, for every workgroup size the
library compiles (one, two, four waves per system) and for inverse Hessians that live in HBM, partly in LDS, or wholly in LDS."""

import shutil
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import ff as orc

ROOT = Path(__file__).resolve().parents[1]
SRC = ROOT / "tests" / "native" / "hess_update_check.hip"


def build(tmp_path_factory, threads):
    if shutil.which("hipcc") is None:
        pytest.skip("needs hipcc")
    exe = tmp_path_factory.getbasetemp() / f"hess_update_check_{threads}"
    if not exe.exists():
        run = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-DCHECK_THREADS={threads}", str(SRC), "-o", str(exe)],
                             capture_output=True, text=True)
        assert run.returncode == 0, run.stderr[-3000:]
    return exe


def test_the_harness_compiles_for_gfx950(tmp_path_factory):
    build(tmp_path_factory, 64)


def random_system(rng, n, identity, aligned=True):
    """generateRandomSystem of the reference's test (tests/test_bfgs_hessian.cpp:86-125)."""
    h = np.eye(n)
    if not identity:
        h = rng.uniform(-1.0, 1.0, size=(n, n))
        h = np.triu(h) + np.triu(h, 1).T
    dgrad, xi, grad = (rng.uniform(-1.0, 1.0, size=n) for _ in range(3))
    xi = np.where((dgrad * xi < 0) == aligned, -xi, xi)
    return h, dgrad, xi, grad


def run(exe, tmp_path, systems, active, lds_kb):
    blob = struct.pack("<i", len(systems))
    for (h, dgrad, xi, grad), on in zip(systems, active):
        blob += struct.pack("<ii", len(dgrad), int(on)) + h.astype("<f8").tobytes() + dgrad.astype("<f8").tobytes() + \
            xi.astype("<f8").tobytes() + grad.astype("<f8").tobytes()
    (tmp_path / "in.bin").write_bytes(blob)
    out = subprocess.run([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(lds_kb)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    data = np.frombuffer((tmp_path / "out.bin").read_bytes(), dtype="<f8")
    got, at = [], 0
    for h, dgrad, _, _ in systems:
        n = len(dgrad)
        got.append((data[at:at + n * n].reshape(n, n), data[at + n * n:at + n * n + n], data[at + n * n + n:at + n * n + 2 * n],
                    data[at + n * n + 2 * n:at + n * n + 3 * n]))
        at += n * n + 3 * n
    assert at == len(data)
    return got


def check(systems, active, got, expect_update=None):
    for (h, dgrad, xi, grad), on, (gh, ghdg, gdg, gxi) in zip(systems, active, got):
        if not on:                                     # an inactive system comes back untouched
            assert np.array_equal(gh, h) and np.array_equal(gdg, dgrad) and np.array_equal(gxi, xi)
            continue
        wh, whdg, wdg, wxi = orc.inverse_hessian_update(h, dgrad, xi, grad)
        if expect_update is not None:
            assert (not np.array_equal(wh, h)) == expect_update
        scale = max(1.0, float(np.abs(wh).max()))
        np.testing.assert_allclose(gh, wh, rtol=0, atol=1e-10 * scale)
        np.testing.assert_allclose(ghdg, whdg, rtol=0, atol=1e-10 * max(1.0, float(np.abs(whdg).max())))
        np.testing.assert_allclose(gdg, wdg, rtol=0, atol=1e-10 * max(1.0, float(np.abs(wdg).max())))
        np.testing.assert_allclose(gxi, wxi, rtol=0, atol=1e-10 * max(1.0, float(np.abs(wxi).max())))
        assert np.array_equal(gh, gh.T)


# workgroup size -> the LDS budgets the library's size classes give it (KiB per workgroup), plus "vectors only" and "everything"
BUDGETS = {64: (0, 19.5, 26, 39.5, 159), 128: (0, 39.5, 79, 159), 256: (0, 79, 159)}


@pytest.mark.gpu
@pytest.mark.parametrize("identity", [True, False])
@pytest.mark.parametrize("dim", [3, 4])
@pytest.mark.parametrize("threads", [64, 128, 256])
def test_single_system(tmp_path_factory, tmp_path, threads, dim, identity):
    """88 atoms (tests/test_bfgs_hessian.cpp:130-202) for the four-wave kernels; the one- and two-wave kernels take the sizes the
    library gives them (up to 232 and 256 coordinates)."""
    atoms = 88 if threads == 256 else (58 if threads == 64 else 64)
    rng = np.random.default_rng(42 + dim + 10 * identity + threads)
    systems = [random_system(rng, atoms * dim if threads == 256 else min(atoms * dim, 232 if threads == 64 else 256), identity)]
    for kb in BUDGETS[threads]:
        check(systems, [True], run(build(tmp_path_factory, threads), tmp_path, systems, [True], kb), expect_update=True)


@pytest.mark.gpu
@pytest.mark.parametrize("identity", [True, False])
@pytest.mark.parametrize("dim", [3, 4])
def test_single_large_system(tmp_path_factory, tmp_path, dim, identity):
    """300 atoms (tests/test_bfgs_hessian.cpp:204-279): 900 / 1200 coordinates, the vectors alone take most of a CU's LDS."""
    rng = np.random.default_rng(7 + dim + 10 * identity)
    systems = [random_system(rng, 300 * dim, identity)]
    for kb in (0, 159):
        check(systems, [True], run(build(tmp_path_factory, 256), tmp_path, systems, [True], kb), expect_update=True)


@pytest.mark.gpu
@pytest.mark.parametrize("identity", [True, False])
@pytest.mark.parametrize("atoms,dim", [(164, 4), (220, 3), (262, 4), (300, 3), (266, 4)])
def test_single_large_system_with_eight_waves(tmp_path_factory, tmp_path, atoms, dim, identity):
    """The pass of the eight-wave class (csrc/minimize.hip: systems of 656 coordinates and more, 512 threads): 656 / 660 / 1048 /
    900 / 1064 coordinates (its eight gradient slabs fit LDS up to 1067), with the vectors alone in LDS and with whatever a whole CU's LDS holds beside them."""
    rng = np.random.default_rng(11 + atoms + dim + 10 * identity)
    systems = [random_system(rng, atoms * dim, identity)]
    for kb in (0, 159):
        check(systems, [True], run(build(tmp_path_factory, 512), tmp_path, systems, [True], kb), expect_update=True)


@pytest.mark.gpu
@pytest.mark.parametrize("identity", [True, False])
@pytest.mark.parametrize("dim", [3, 4])
@pytest.mark.parametrize("threads,third", [(64, 33), (128, 33), (256, 33), (256, 300)])
def test_several_systems_one_inactive(tmp_path_factory, tmp_path, threads, third, dim, identity):
    """{3, 2, 33, 14} and {3, 2, 300, 14} atoms with the third system skipped (tests/test_bfgs_hessian.cpp:281-485)."""
    rng = np.random.default_rng(100 + dim + 10 * identity + third)
    systems = [random_system(rng, a * dim, identity) for a in (3, 2, third, 14)]
    active = [True, True, False, True]
    for kb in (BUDGETS[threads] if third == 33 else (0, 159)):
        check(systems, active, run(build(tmp_path_factory, threads), tmp_path, systems, active, kb))
    everyone = [True] * 4
    if third == 33:
        check(systems, everyone, run(build(tmp_path_factory, threads), tmp_path, systems, everyone, BUDGETS[threads][1]), expect_update=True)


@pytest.mark.gpu
@pytest.mark.parametrize("identity", [True, False])
@pytest.mark.parametrize("dim", [3, 4])
@pytest.mark.parametrize("threads", [64, 128, 256])
def test_update_skipped_when_the_signs_disagree(tmp_path_factory, tmp_path, threads, dim, identity):
    """dGrad . xi < 0: the matrix and dGrad stay, only the direction -H grad is formed (tests/test_bfgs_hessian.cpp:487-585)."""
    rng = np.random.default_rng(200 + dim + 10 * identity)
    systems = [random_system(rng, a * dim, identity, aligned=False) for a in (3, 2, 33, 14)]
    got = run(build(tmp_path_factory, threads), tmp_path, systems, [True] * 4, BUDGETS[threads][1])
    check(systems, [True] * 4, got, expect_update=False)
    for (h, dgrad, _, _), (gh, _, gdg, _) in zip(systems, got):
        assert np.array_equal(gh, h) and np.array_equal(gdg, dgrad)
