"""bench.py's contract with the driver: exactly ONE line on stdout, valid JSON with the agreed keys, for the plain
run and for the N > 1 code path (process group + bucketed all-gather, exercised on one rank with --force-dist)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


@pytest.mark.parametrize("extra", [[], ["--force-dist"], ["--workload", "urdf_panda"]])
def test_one_json_line(extra):
    env = dict(os.environ, MASTER_PORT="29547")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3",
                        "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert KEYS <= set(d), sorted(KEYS - set(d))
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["value"] > 0 and d["unit"] == "M evals/s" and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(rf)
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3


def test_cpu_baseline_leg():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--batch", "4096"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip())
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["value"] > 0
