"""bench.py's contract with the driver: exactly ONE line on stdout, valid JSON with the agreed keys, for every BASELINE
workload, both scaling modes and the N > 1 code path (process group, per-call all-gather, sharded fused Adam and the
`variants` object, exercised on one rank with --force-dist)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _run(extra, steps=10, warmup=3, timeout=600):
    env = dict(os.environ, MASTER_PORT="29547")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warmup)] + extra,
                       capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [[], ["--workload", "cfg2"], ["--workload", "cfg3"], ["--workload", "cfg4"],
                                   ["--workload", "cfg5"], ["--scaling", "strong"]],
                         ids=["headline", "cfg2", "cfg3", "cfg4", "cfg5", "headline-strong"])
def test_one_json_line_per_baseline_workload(extra):
    d = _run(["--no-cpu-baseline"] + extra)
    assert KEYS <= set(d), sorted(KEYS - set(d))
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["value"] > 0 and d["unit"] == "M evals/s" and d["dtype"] == "f32"
    assert d["scaling"] == ("strong" if "strong" in extra else "weak")
    cfg = d["config"]
    assert "workload" in cfg and "model" not in cfg and cfg["global_batch"] == cfg["batch_per_gpu"]
    rf = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "mfma"} <= set(rf)
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0 < rf["frac"] < 1
    assert rf["kernel_ms"] <= d["ms_per_step"] * 1.05
    mf = rf["mfma"]
    assert mf["used"] is False and mf["instructions_per_launch"] == 0 and mf["measured_variant"]["busy_frac"] > 0
    if rf["traffic"] is not None:
        assert "profiles/" in rf["traffic_source"]


@pytest.mark.parametrize("extra", [[], ["--workload", "cfg3", "--scaling", "strong"], ["--workload", "cfg5", "--scaling", "strong"],
                                   ["--gather", "serial"], ["--gather", "bucketed", "--no-variants"], ["--no-gather", "--no-variants"]],
                         ids=["headline", "cfg3-strong", "cfg5-strong", "serial", "bucketed", "none"])
def test_distributed_code_path_on_one_rank(extra):
    d = _run(["--no-cpu-baseline", "--force-dist"] + extra, steps=24, warmup=4)
    assert d["n_gpus"] == 1 and d["value"] > 0
    m = d["multi"]
    assert m["ranks"] == 1 and "nccl" in m["backend"]
    if "cfg5" in extra:
        assert m["gather"] == "summaries" and d["config"]["global_batch"] == 256 * 50
    elif "--no-gather" in extra:
        assert m["gather"] == "none" and m["gather_ms"] is None
    else:
        assert m["gather_ms"] is not None and m["gather_ms"] > 0
        assert m["gather"] == (extra[extra.index("--gather") + 1] if "--gather" in extra else "per-call")
    if "--no-variants" not in extra:
        v = d["variants"]
        assert v["other_scaling"]["scaling"] == ("weak" if "strong" in extra else "strong") and v["other_scaling"]["value"] > 0
        if "cfg5" not in extra:
            assert {k for k in v if k.startswith("gather_")} == {f"gather_{g}" for g in ("per-call", "serial", "bucketed", "none")
                                                                  if g != m["gather"]}
    if "cfg3" in extra:
        assert d["config"]["global_batch"] == 65536 and d["scaling"] == "strong"


def test_cpu_baseline_leg():
    d = _run(["--batch", "4096"], steps=5, warmup=2)
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["value"] > 0
