"""UFF conformer optimisation (reference API: nvmolkit/uffOptimization.py).

Not built yet: the UFF term kernels (reference src/forcefields/uff_kernels_device.cuh) are the
next force-field family after DG / ETK / MMFF; the BFGS driver and the flattened-term ABI they plug into exist
(nvmolkit_amd/forcefield.py).  The entry point fails loudly instead of falling back to a CPU path."""

from __future__ import annotations


def UFFOptimizeMoleculesConfs(molecules, maxIters: int = 1000, vdwThreshold: float = 10.0,
                              ignoreInterfragInteractions: bool = True, hardwareOptions=None, output=None, targetGpu: int = -1):
    raise NotImplementedError("UFF terms are not implemented in this build (SURVEY.md §8 row F2); "
                              "MMFFOptimizeMoleculesConfs / forcefield.FlatForcefieldBatch cover MMFF94")
