"""nvmolkit_amd — MI355X (gfx950) implementation of nvMolKit's batched hot path.

Same public modules as the reference's ``nvmolkit`` package for the path in scope:
``fingerprints``, ``similarity``, ``clustering``, ``embedMolecules``, ``mmffOptimization``,
``uffOptimization``, ``types``.  The compute lives in ``lib/libnvmolkit_amd.so`` (hand-written HIP,
C ABI in ``include/nvmolkit_amd.h``); torch is used for device memory, streams and
``torch.distributed`` only.
"""

__version__ = "0.1.0"
