"""nvmolkit_amd/csrc/ff_grad.h (the hand-derived angular gradients the fused BFGS kernels call) compiled for the HOST and
checked, term group by term group, against the C oracle's gradient of a system that holds only that group — no GPU needed.
The GPU tests then check the kernels as a whole (tests/test_forcefield_gpu.py, tests/test_bfgs_parity_gpu.py)."""

import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest

from nvmolkit_amd import synthetic
from oracle import ff as off
from oracle import ffc

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("ffgrad") / "libffgrad.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", str(ROOT / "tests" / "native" / "ff_grad_host.cpp"), "-o", str(out)],
                   check=True)
    lib = ctypes.CDLL(str(out))
    lib.chk_term_gradient.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    lib.chk_term_gradient.restype = None
    return lib


# (check code, force-field kind, term group) of every angular group
CASES = [(0, off.DG, 1), (1, off.ETK, 0), (2, off.ETK, 1), (3, off.ETK, 4), (4, off.MMFF, 1), (5, off.MMFF, 2), (6, off.MMFF, 3),
         (7, off.MMFF, 4), (8, off.UFF, 1), (9, off.UFF, 2), (10, off.UFF, 3)]


@pytest.mark.parametrize("code,kind,group", CASES)
@pytest.mark.parametrize("n_atoms", [6, 23, 60])
def test_term_group_gradient_equals_oracle(host_lib, code, kind, group, n_atoms):
    rng = np.random.default_rng(100 * code + n_atoms)
    pos, groups = synthetic.random_ff_system(kind, n_atoms, rng)
    layout = off.LAYOUT[kind]
    only = [(g if k == group else (np.zeros((0, layout[k][0]), np.int64), np.zeros((0, layout[k][1])))) for k, g in enumerate(groups)]
    a_s, flat, stacked = synthetic.build_ff_batch_arrays(kind, [(pos, only)])
    w0 = 0.7
    want = ffc.Batch(kind, a_s, stacked).gradient(flat, w0, 0.3)
    idx = np.ascontiguousarray(groups[group][0], dtype=np.int32)
    par = np.ascontiguousarray(groups[group][1], dtype=np.float64)
    assert len(idx) > 0
    got = np.zeros_like(flat)
    p = np.ascontiguousarray(flat)
    host_lib.chk_term_gradient(code, off.DIM[kind], p.ctypes.data, idx.ctypes.data, len(idx), idx.shape[1], par.ctypes.data,
                               par.shape[1] if par.size else 0, w0, got.ctypes.data)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(want).max()))
    assert np.abs(want).max() > 0.0


def test_degenerate_geometries_give_no_gradient(host_lib):
    """Zero-length arms, collinear dihedrals and planar sine floors: no NaN, no gradient (as the dual-number path)."""
    pos = np.zeros((4, 3))
    pos[1] = (1.0, 0.0, 0.0)
    pos[2] = (2.0, 0.0, 0.0)
    pos[3] = (3.0, 0.0, 0.0)                                   # four collinear atoms
    idx4 = np.array([[0, 1, 2, 3]], dtype=np.int32)
    idx3 = np.array([[0, 0, 2]], dtype=np.int32)               # a zero-length arm
    for code, idx, par in ((1, idx4, np.r_[np.ones(6), np.ones(6)]), (7, idx4, np.ones(3)), (9, idx4, np.array([1.0, 3.0, 1.0])),
                           (6, idx4, np.ones(1)), (2, idx4, np.array([1.0, -1.0, 0.0, 10.0])), (4, idx3, np.array([109.0, 1.0, 0.0])),
                           (5, idx3, np.array([109.0, 1.5, 1.5, 0.1, 0.1])), (8, idx3, np.array([1.9, 100.0, 3.0, 0.0, 0.0, 0.0]))):
        got = np.zeros(12)
        par = np.ascontiguousarray(par.reshape(1, -1))
        host_lib.chk_term_gradient(code, 3, pos.ctypes.data, idx.ctypes.data, 1, idx.shape[1], par.ctypes.data, par.shape[1], 1.0,
                                   got.ctypes.data)
        assert np.all(np.isfinite(got)) and not got.any(), code
