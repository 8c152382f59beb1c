"""The Python interface of the hot path against the reference's: every public function, class, method and property of the
reference's modules that are in scope must exist in nvmolkit_amd under the same name, take the
reference's parameters in the reference's order (extra trailing parameters are allowed) and have the same defaults.
The inventory (tests/golden/reference_python_api.json) was read from the reference's sources with `ast` by
tests/golden/make_api_fixture.py; the reference itself is not needed to run this test."""

import enum
import importlib
import inspect
import json
from pathlib import Path

import pytest

API = json.loads((Path(__file__).parent / "golden" / "reference_python_api.json").read_text())
# out of scope: substructure search and TFD are other features; the two underscore modules are
# private helpers of the reference's implementation (Triton kernels of fused Butina, the RDKit <-> C++ MMFF property bridge)
OUT_OF_SCOPE = {"substructure", "tfd", "_fusedButina", "_mmff_bridge"}
IN_SCOPE = sorted(set(API) - OUT_OF_SCOPE)


def test_every_in_scope_module_exists():
    assert IN_SCOPE == ["batchedForcefield", "clustering", "conformerRmsd", "embedMolecules", "fingerprints", "mmffOptimization",
                        "similarity", "types", "uffOptimization"]
    for name in IN_SCOPE:
        importlib.import_module(f"nvmolkit_amd.{name}")


def _normalise(text):
    """Default values compare as text; module prefixes and quote style do not matter."""
    if text is None:
        return None
    return text.replace('"', "'").replace("nvmolkit.types.", "").replace("types.", "").replace("float('nan')", "nan")


def _default_text(value):
    if value is inspect.Parameter.empty:
        return None
    if isinstance(value, enum.Enum):
        return f"{type(value).__name__}.{value.name}"
    return repr(value)


def _check_params(where, want, fn):
    sig = inspect.signature(fn)
    have = [p for p in sig.parameters.values() if p.name not in ("self", "cls")]
    names = [p.name for p in have]
    var_keyword = any(p.kind is inspect.Parameter.VAR_KEYWORD for p in have)
    positional = [w for w in want if w["kind"] == "positional"]
    assert names[:len(positional)] == [w["name"] for w in positional], f"{where}: positional parameters {names} vs reference {positional}"
    by_name = {p.name: p for p in have}
    for w in want:
        if w["name"] not in by_name:
            assert var_keyword, f"{where}: parameter {w['name']} is missing"
            continue
        p = by_name[w["name"]]
        if w["kind"] == "positional":
            assert p.kind in (inspect.Parameter.POSITIONAL_OR_KEYWORD, inspect.Parameter.POSITIONAL_ONLY), f"{where}: {w['name']} must be positional"
        got, ref = _normalise(_default_text(p.default)), _normalise(w["default"])
        if (ref is None) != (got is None):
            # a parameter the reference requires may be optional here, never the other way round
            assert ref is None or got is not None, f"{where}: {w['name']} has default {ref} in the reference and none here"
            continue
        if ref is not None and where + "." + w["name"] not in DOCUMENTED_DEFAULT_DIFFERENCES:
            assert got == ref, f"{where}: default of {w['name']} is {got}, reference {ref}"
    # parameters added here must not get in the way of a reference-style call
    for p in have[len(positional):]:
        if p.name not in {w["name"] for w in want} and p.kind not in (inspect.Parameter.VAR_KEYWORD, inspect.Parameter.VAR_POSITIONAL):
            assert p.default is not inspect.Parameter.empty, f"{where}: extra parameter {p.name} has no default"


# "maxIters: None" here stands for the reference's per-force-field default (200 MMFF / 1000 UFF) in the shared base class
DOCUMENTED_DEFAULT_DIFFERENCES = {"batchedForcefield.MMFFBatchedForcefield.minimize.maxIters",
                                  "batchedForcefield.UFFBatchedForcefield.minimize.maxIters"}


@pytest.mark.parametrize("module", IN_SCOPE)
def test_public_names_and_signatures(module):
    mod = importlib.import_module(f"nvmolkit_amd.{module}")
    for name, spec in API[module].items():
        assert hasattr(mod, name), f"nvmolkit_amd.{module}.{name} is missing"
        obj = getattr(mod, name)
        if spec["type"] == "function":
            _check_params(f"{module}.{name}", spec["params"], obj)
            continue
        assert inspect.isclass(obj), f"{module}.{name} is a class in the reference"
        for mname, mspec in spec["methods"].items():
            assert hasattr(obj, mname), f"{module}.{name}.{mname} is missing"
            member = inspect.getattr_static(obj, mname)
            if mspec.get("property"):
                assert isinstance(member, property) or not callable(member), f"{module}.{name}.{mname} is a property in the reference"
                continue
            if mname == "__init__" and issubclass(obj, enum.Enum):
                continue
            _check_params(f"{module}.{name}.{mname}", mspec["params"], getattr(obj, mname))


def test_minimize_defaults_per_force_field():
    """200 iterations for MMFF, 1000 for UFF (reference nvmolkit/batchedForcefield.py:565-571, 681-687)."""
    import nvmolkit_amd.batchedForcefield as bff
    src = inspect.getsource(bff.FlatBatchedForcefield.minimize)
    assert "200 if self.kind == MMFF else 1000" in src
    want = {m["name"]: m["default"] for m in API["batchedForcefield"]["UFFBatchedForcefield"]["methods"]["minimize"]["params"]}
    assert want["maxIters"] == "1000" and want["target_gpu"] == "None"
    assert {m["name"]: m["default"] for m in API["batchedForcefield"]["MMFFBatchedForcefield"]["methods"]["minimize"]["params"]}["maxIters"] == "200"
