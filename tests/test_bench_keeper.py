"""bench.py's LineKeeper on the CPU: the ONE stdout line survives the death of the process that measured it
(an abort from a library thread inside a side measurement must not lose the SCALE record)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import os, sys
sys.path.insert(0, {root!r})
import bench
k = bench.LineKeeper(os.dup(1))
k.primary({{"value": 1, "n_gpus": 8}})
mode = sys.argv[1]
if mode == "abort":
    os.abort()                      # what c10d's watchdog does to the process: no Python clean-up runs
if mode == "kill":
    os.kill(os.getpid(), 9)
k.final({{"value": 1, "n_gpus": 8, "variants": {{"x": 2}}}})
if mode == "abort_after_final":
    os.abort()
k.close()
"""


def _run(mode):
    r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT), mode], capture_output=True, text=True, timeout=300, cwd=ROOT)
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, (r.stdout, r.stderr[-500:])
    return r.returncode, json.loads(lines[0])


def test_primary_line_survives_abort_and_kill():
    for mode in ("abort", "kill"):
        rc, d = _run(mode)
        assert rc != 0 and d == {"value": 1, "n_gpus": 8}


def test_final_line_replaces_the_primary_one():
    rc, d = _run("clean")
    assert rc == 0 and d["variants"] == {"x": 2}
    rc, d = _run("abort_after_final")
    assert rc != 0 and d["variants"] == {"x": 2}


def test_bare_gpus_n_becomes_its_own_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE around it re-executes itself under torch.distributed.run (VERDICT r4
    item 1).  Without a GPU the two ranks it starts stop at `require_gpu` - loudly, no CPU fallback - which is enough to see
    here that BOTH exist; the measured form of this test is tests/test_gpu_bench_contract.py."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the GPU form of this test runs the whole job")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "local_rank: 0" in r.stderr and "local_rank: 1" in r.stderr, r.stderr[-1500:]
    assert "no CPU fallback" in r.stderr
