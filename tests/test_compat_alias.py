"""compat/nvmolkit: the reference's package name resolves to the modules of nvmolkit_amd (code written against
``nvmolkit.similarity`` / ``nvmolkit.clustering`` / ... runs unchanged with compat/ on the path)."""

import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_reference_package_name_resolves_to_this_library():
    code = (
        "import sys; sys.path[:0] = [r'%s', r'%s']\n"
        "import nvmolkit\n"
        "assert 'torch' not in sys.modules, 'import nvmolkit alone must stay light'\n"
        "from nvmolkit.similarity import crossTanimotoSimilarity\n"
        "import nvmolkit.clustering as c, nvmolkit_amd.clustering as d, nvmolkit_amd.similarity as s\n"
        "assert c is d and crossTanimotoSimilarity is s.crossTanimotoSimilarity\n"
        "assert nvmolkit.types.AsyncGpuResult is __import__('nvmolkit_amd.types', fromlist=['x']).AsyncGpuResult\n"
        "from nvmolkit.clustering import butina, fused_butina\n"
        "import importlib\n"
        "assert s.__spec__.name == 'nvmolkit_amd.similarity' and s.__spec__.origin.endswith('similarity.py'), s.__spec__\n"
        "assert importlib.reload(s) is s and c.__spec__.name == 'nvmolkit_amd.clustering'\n"
        "try:\n"
        "    import nvmolkit.substructure\n"
        "    raise SystemExit('a module outside the hot path imported')\n"
        "except ImportError:\n"
        "    pass\n"
        "print('alias ok')\n" % (ROOT, ROOT / "compat"))
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "alias ok" in run.stdout, run.stdout + run.stderr
