"""Device3DResult container semantics (reference: nvmolkit/types.py:197-319; nvmolkit/tests/test_types.py) on CPU
tensors — the GPU chaining tests live in tests/test_device_chain_gpu.py."""

import numpy as np
import pytest
import torch

from nvmolkit_amd.types import AsyncGpuResult, CoordinateOutput, Device3DResult


def make(n_mols=4):
    # molecule 0: 1 conformer of 2 atoms; molecule 2: 2 conformers of 3 and 2 atoms; molecules 1 and 3: none
    values = torch.arange(7 * 3, dtype=torch.float64).reshape(7, 3)
    return Device3DResult(values, torch.tensor([0, 2, 5, 7], dtype=torch.int32), torch.tensor([0, 2, 2], dtype=torch.int32),
                          torch.tensor([0, 0, 1], dtype=torch.int32), gpu_id=0, n_mols=n_mols,
                          energies=torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64),
                          converged=torch.tensor([1, 0, 1], dtype=torch.int8)), values


def test_fields_and_per_molecule_views_share_storage():
    res, values = make()
    assert res.num_conformers == 3 and res.n_mols == 4 and res.gpu_id == 0
    assert all(isinstance(x, AsyncGpuResult) for x in (res.values, res.atom_starts, res.mol_indices, res.conf_indices,
                                                       res.energies, res.converged))
    per = res.per_molecule()
    assert [len(p) for p in per] == [1, 0, 2, 0]
    assert per[0][0].shape == (2, 3) and per[2][0].shape == (3, 3) and per[2][1].shape == (2, 3)
    assert torch.equal(per[2][1], values[5:7])
    per[2][1][0, 0] = -1.0                      # a view, not a copy
    assert values[5, 0] == -1.0


def test_dense_padding_and_masks():
    res, values = make()
    d = res.dense()
    assert d.values.shape == (4, 2, 3, 3) and d.conf_mask.shape == (4, 2) and d.atom_mask.shape == (4, 2, 3)
    assert d.conf_mask.tolist() == [[True, False], [False, False], [True, True], [False, False]]
    assert int(d.atom_mask.sum()) == 7
    assert torch.equal(d.values[0, 0, :2], values[0:2]) and torch.equal(d.values[2, 0], values[2:5])
    assert torch.equal(d.values[2, 1, :2], values[5:7])
    assert torch.isnan(d.values[~d.atom_mask]).all()
    assert torch.equal(d.values[d.atom_mask], values)         # row order of the flat form is preserved
    z = res.dense(pad_value=0.0)
    assert float(z.values[1].abs().sum()) == 0.0


def test_empty_result_and_validation():
    e = Device3DResult(torch.zeros((0, 3), dtype=torch.float64), torch.zeros(1, dtype=torch.int32),
                       torch.zeros(0, dtype=torch.int32), torch.zeros(0, dtype=torch.int32), 0, 3)
    d = e.dense()
    assert d.values.shape == (3, 0, 0, 3) and d.conf_mask.shape == (3, 0) and d.atom_mask.shape == (3, 0, 0)
    assert e.per_molecule() == [[], [], []] and e.num_conformers == 0
    with pytest.raises(ValueError):
        Device3DResult(torch.zeros((2, 3), dtype=torch.float64), torch.tensor([0, 2], dtype=torch.int32),
                       torch.tensor([0, 1], dtype=torch.int32), torch.tensor([0], dtype=torch.int32), 0, 2)
    assert CoordinateOutput("device") is CoordinateOutput.DEVICE and CoordinateOutput("rdkit") is CoordinateOutput.RDKIT_CONFORMERS


def test_optimizer_entry_points_validate_like_the_reference():
    """Argument / error behaviour that needs neither RDKit nor a GPU (nvmolkit/mmffOptimization.py:125-162,
    uffOptimization.py:88-113)."""
    from nvmolkit_amd.mmffOptimization import MMFFOptimizeMoleculesConfs
    from nvmolkit_amd.uffOptimization import UFFOptimizeMoleculesConfs

    assert MMFFOptimizeMoleculesConfs([]) == [] and UFFOptimizeMoleculesConfs([]) == []
    for fn in (MMFFOptimizeMoleculesConfs, UFFOptimizeMoleculesConfs):
        with pytest.raises(ValueError, match="requires at least one molecule"):
            fn([], output=CoordinateOutput.DEVICE)


def test_uff_torsion_and_angle_shape_rules():
    """Pure-Python parts of the UFF flattener (rdkit_extensions/uff_flattened_builder.cpp:62-70, 86-136)."""
    from nvmolkit_amd.uffOptimization import _angle_coefficients, _torsion_shape

    c0, c1, c2 = _angle_coefficients(np.deg2rad(109.47))
    th = np.deg2rad(109.47)
    # the general bend has its minimum (value 0, slope 0) at theta0
    assert c0 + c1 * np.cos(th) + c2 * np.cos(2 * th) == pytest.approx(0.0, abs=1e-12)
    assert -c1 * np.sin(th) - 2 * c2 * np.sin(2 * th) == pytest.approx(0.0, abs=1e-12)
    assert _torsion_shape(1.0, 6, 6, True, True, False) == (3, -1.0)       # sp3-sp3
    assert _torsion_shape(1.0, 8, 16, True, True, False) == (2, -1.0)      # group-6 single bond
    assert _torsion_shape(2.0, 6, 6, False, False, False) == (2, 1.0)      # sp2-sp2
    assert _torsion_shape(1.0, 8, 6, True, False, False) == (2, -1.0)      # sp3 group-6 next to sp2 non-group-6
    assert _torsion_shape(1.0, 6, 6, True, False, True) == (3, -1.0)       # propene-like
    assert _torsion_shape(1.0, 6, 6, True, False, False) == (6, 1.0)


# ---- the scenarios of the reference's own Device3DResult tests (nvmolkit/tests/test_types.py:105-296), on CPU tensors: values
# encode (conformer, atom, axis) as conformer * 1000 + atom * 10 + axis, so every view can be checked by value.

def coded(atom_counts, mol_indices, n_mols, conf_indices=None):
    starts = np.concatenate([[0], np.cumsum(atom_counts)]).astype(np.int64)
    values = torch.empty((int(starts[-1]), 3), dtype=torch.float64)
    for c, k in enumerate(atom_counts):
        for a in range(k):
            values[starts[c] + a] = torch.tensor([c * 1000 + a * 10 + x for x in range(3)], dtype=torch.float64)
    if conf_indices is None:
        seen, conf_indices = {}, []
        for m in mol_indices:
            conf_indices.append(seen.get(m, 0))
            seen[m] = conf_indices[-1] + 1
    return Device3DResult(values, torch.tensor(starts, dtype=torch.int32), torch.tensor(mol_indices, dtype=torch.int32),
                          torch.tensor(conf_indices, dtype=torch.int32), gpu_id=0, n_mols=n_mols)


def test_reference_scenarios_per_molecule():
    assert coded([3, 5, 2], [0, 0, 1], 2).num_conformers == 3                                   # :105-111
    nested = coded([2, 3, 4], [1, 0, 1], 2).per_molecule()                                       # :114-139
    assert [len(x) for x in nested] == [1, 2]
    assert nested[0][0].shape == (3, 3) and nested[1][0].shape == (2, 3) and nested[1][1].shape == (4, 3)
    assert nested[1][0][0, 0] == 0 and nested[0][0][0, 0] == 1000 and nested[1][1][0, 0] == 2000 and nested[0][0][2, 2] == 1022
    nested = coded([2], [0], 4).per_molecule()                                                   # :142-154
    assert len(nested) == 4 and len(nested[0]) == 1 and nested[1] == [] and nested[2] == [] and nested[3] == []
    res = coded([3], [0], 1)                                                                     # :157-167
    res.per_molecule()[0][0][0, 0] = -7.0
    assert res.values.torch()[0, 0] == -7.0


def test_reference_scenarios_dense():
    from nvmolkit_amd.types import Dense3DResult

    out = coded([2, 4, 3], [0, 1, 2], 3).dense()                                                 # :170-184
    assert isinstance(out, Dense3DResult)
    assert out.values.shape == (3, 1, 4, 3) and out.conf_mask.shape == (3, 1) and out.atom_mask.shape == (3, 1, 4)
    assert out.values.dtype == torch.float64 and out.conf_mask.dtype == torch.bool and out.atom_mask.dtype == torch.bool
    counts = [5, 3, 2]                                                                           # :187-233
    atom_counts, mol_indices, conf_indices = [], [], []
    for m, k in enumerate(counts):
        for slot in range(k):
            atom_counts.append(2 + slot % 3)
            mol_indices.append(m)
            conf_indices.append(slot)
    out = coded(atom_counts, mol_indices, 3, conf_indices).dense()
    assert out.values.shape == (3, 5, 4, 3) and out.conf_mask.sum(dim=1).tolist() == counts
    assert out.atom_mask.sum(dim=(1, 2)).tolist() == [sum(atom_counts[:5]), sum(atom_counts[5:8]), sum(atom_counts[8:])]
    assert out.values[0, 4, 0, 0] == 4000 and out.values[1, 2, 0, 0] == 7000 and out.values[2, 1, 0, 0] == 9000
    assert torch.isnan(out.values[1, 4, 0, 0]) and torch.isnan(out.values[2, 2, 0, 0])
    out = coded([3, 2], [0, 2], 3).dense()                                                       # :236-249
    assert out.conf_mask[0].any() and not out.conf_mask[1].any() and out.conf_mask[2].any()
    assert torch.isnan(out.values[1]).all() and not out.atom_mask[1].any()
    out = coded([2, 4], [0, 0], 1).dense()                                                       # :252-266
    assert out.values.shape == (1, 2, 4, 3) and out.conf_mask[0].tolist() == [True, True]
    assert out.atom_mask[0, 0].tolist() == [True, True, False, False] and out.atom_mask[0, 1].tolist() == [True] * 4
    assert torch.isnan(out.values[0, 0, 2, 0]) and not torch.isnan(out.values[0, 0, 1, 0])
    out = coded([1, 3], [0, 1], 2).dense(pad_value=-1.0)                                         # :269-280
    assert out.values.shape == (2, 1, 3, 3)
    assert out.values[0, 0, 1, 0] == -1.0 and out.values[0, 0, 2, 2] == -1.0 and out.values[1, 0, 2, 2] == 1022
    out = coded([], [], 3).dense()                                                               # :283-296
    assert out.values.shape == (3, 0, 0, 3) and out.conf_mask.shape == (3, 0) and out.atom_mask.shape == (3, 0, 0)
    assert out.values.dtype == torch.float64


@pytest.mark.parametrize("invalid", [0, -2, -99])
def test_hardware_options_refuse_invalid_batches_per_gpu(invalid):
    """nvmolkit/tests/test_types.py:30-38: at construction and through the setter, with the reference's message."""
    from nvmolkit_amd.types import HardwareOptions

    with pytest.raises(ValueError, match="batchesPerGpu must be greater than 0 or -1"):
        HardwareOptions(batchesPerGpu=invalid)
    hw = HardwareOptions()
    with pytest.raises(ValueError, match="batchesPerGpu must be greater than 0 or -1"):
        hw.batchesPerGpu = invalid
