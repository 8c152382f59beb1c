"""The bench.py contract, as far as it can be held without a GPU: the flags the driver passes exist with defaults that finish in
minutes, and the last bench line committed under profiles/ (written by bench.py on an MI355X) carries every field the contract
names — metric / config of BASELINE.json, the roofline object and the cpu_baseline object."""

import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_flags_and_defaults():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1000:]
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
    src = (ROOT / "bench.py").read_text()
    assert 'add_argument("--gpus", type=int, default=1)' in src
    assert 'add_argument("--steps", type=int, default=3)' in src and 'add_argument("--warmup", type=int, default=1)' in src


def test_committed_bench_line_has_every_field_of_the_contract():
    line = json.loads((ROOT / "profiles" / "r03_head_bench.json").read_text().strip().splitlines()[-1])
    baseline = json.loads((ROOT / "BASELINE.json").read_text())
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["scaling"] == "weak" and line["vs_baseline"] is None and "synthetic" in line["data"]
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["unit"] == "pairs/s" and "Tanimoto" in line["metric"]
    assert any(word in json.dumps(baseline) for word in ("Tanimoto", "tanimoto"))
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0.5 < roof["frac"] < 1.0
    assert roof["traffic"] is None or roof["traffic"] >= 0.9 * roof["algorithmic_bytes_per_full_launch"]
    # value, ms_per_step and the roofline agree with each other: pairs per step / time, bytes per step / time
    pairs = 1.0e6 * 1.0e6
    assert abs(line["value"] - pairs / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["unit"] == line["unit"]
    for name in ("fused_butina", "conformers"):
        block = line["secondary"][name]
        assert "roofline" in block and "cpu_baseline" in block and block["roofline"]["frac"] > 0.0
