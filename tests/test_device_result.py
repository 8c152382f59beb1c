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
