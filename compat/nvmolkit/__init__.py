"""``nvmolkit`` as a name for :mod:`nvmolkit_amd` — for code written against the reference's package.

Put this directory on ``PYTHONPATH`` (``PYTHONPATH=/path/to/repo:/path/to/repo/compat``) and
``from nvmolkit.similarity import crossTanimotoSimilarity``, ``import nvmolkit.clustering`` ... resolve to the modules of the same
name in ``nvmolkit_amd`` (imported on first use, so ``import nvmolkit`` alone loads nothing heavy).  Only the modules of the
hot path exist: ``substructure``, ``tfd`` and ``autotune`` raise ``ImportError`` as any missing module."""

import importlib
import importlib.abc
import importlib.util
import sys

_IMPL = "nvmolkit_amd"
_MODULES = ("similarity", "fingerprints", "clustering", "embedMolecules", "mmffOptimization", "uffOptimization", "conformerRmsd",
            "batchedForcefield", "types")


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        pkg, _, leaf = fullname.rpartition(".")
        if pkg == __name__ and leaf in _MODULES:
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        # importlib stamps the alias spec onto whatever create_module returns; the implementation module must keep its own
        # (importlib.reload and tooling read __spec__ / __loader__), so both are put back in exec_module
        impl = importlib.import_module(_IMPL + "." + spec.name.rpartition(".")[2])
        self._own[impl.__name__] = (impl.__spec__, getattr(impl, "__loader__", None))
        return impl

    def exec_module(self, module):  # the implementation module is already initialised
        spec, loader = self._own.get(module.__name__, (None, None))
        if spec is not None:
            module.__spec__ = spec
            module.__loader__ = loader

    _own = {}


sys.meta_path.insert(0, _AliasFinder())


def __getattr__(name):
    if name in _MODULES:
        return importlib.import_module(__name__ + "." + name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
