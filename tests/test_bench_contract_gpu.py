"""GPU: bench.py prints ONE JSON line with the fields the driver and the judge read (metric contract)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2"] + list(flags),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_default_line_has_the_contract_fields():
    j = _run("--cpu-seconds", "3")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 8 and j["warmup"] == 2
    assert j["unit"] == "particle-steps/s" and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["vs_baseline"] is None and j["dtype"] == "f64" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - 4096 * 32 * 8 / (j["ms_per_step"] * 8e-3)) / j["value"] < 1e-6
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert r["traffic"] is None or r["traffic"] > 0
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == j["unit"] and c["sample"]
    assert j["solver_failures"] == 0 and j["final_distance_to_target"] < 1.0          # (10 steps: still on its way)


@pytest.mark.parametrize("flags", [("--dtype", "f32"), ("--noise", "mt19937"), ("--no-graph",), ("--particles", "1024")])
def test_variants_run(flags):
    j = _run("--no-cpu-baseline", *flags)
    assert j["value"] > 0 and j.get("cpu_baseline") is None
    assert j["solver_failures"] == 0
