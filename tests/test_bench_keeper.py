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
