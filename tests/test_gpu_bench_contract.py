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
    d = _run(["--no-cpu-baseline", "--no-configs"] + extra)
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
                                   ["--gather", "overlapped"], ["--gather", "bucketed", "--no-variants"], ["--no-gather", "--no-variants"]],
                         ids=["headline", "cfg3-strong", "cfg5-strong", "overlapped", "bucketed", "none"])
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
        # the default is the in-order per-call gather (always works): it is measured and parked first (`multi.primary`); the
        # line's value is the fastest completed form that delivers every call's scores (`multi.gather`, promotion)
        asked = extra[extra.index("--gather") + 1] if "--gather" in extra else "per-call"
        assert m["primary"]["gather"] == asked and m["primary"]["value"] > 0
        if asked in ("per-call", "overlapped", "graph") and "--no-variants" not in extra:
            assert m["gather"] in ("per-call", "overlapped", "graph") and d["value"] >= m["primary"]["value"]
            assert m["promoted"] == (m["gather"] != asked)
        else:
            assert m["gather"] == asked and not m["promoted"]
    if "--no-variants" not in extra:
        v = d["variants"]
        assert v["other_scaling"]["scaling"] == ("weak" if "strong" in extra else "strong") and v["other_scaling"]["value"] > 0
        if "cfg5" not in extra:
            assert {k for k in v if k.startswith("gather_")} == {f"gather_{g}" for g in ("graph", "per-call", "overlapped", "bucketed", "none")
                                                                  if g != m["gather"]}
            for g in ("graph", "per-call", "overlapped"):   # the candidates are timed like the primary line: exactly K steps
                if g != m["gather"] and "error" not in v[f"gather_{g}"]:
                    assert v[f"gather_{g}"]["steps"] == d["steps"]
            assert "gather_exposed_ms" in m
    if "cfg3" in extra:
        assert d["config"]["global_batch"] == 65536 and d["scaling"] == "strong"


def test_headline_line_carries_every_baseline_config():
    """the driver runs `python bench.py` only: its one line must hold a short measurement of every BASELINE.json config
    that runs on a GPU"""
    d = _run(["--no-cpu-baseline"], steps=10, warmup=3, timeout=900)
    cf = d["configs"]
    assert set(cf) == {"cfg2", "cfg2_panda", "cfg3", "cfg3_b65536", "cfg3_poly", "cfg4", "cfg5", "cfg5_shard32", "cfg5_c5", "headline_rq"}
    assert all("error" not in v for v in cf.values()), cf
    assert cf["cfg5_shard32"]["batch"] == 32 * 50 and cf["cfg3_b65536"]["batch"] == 65536
    # config #5's loop on config #3's five-class model (round 6): two sweeps per iteration inside the persistent launch - more than
    # one sweep's time, far below the Python loop's (538 us per iteration on the single-class problem)
    # (2.0 - 2.1x when the GPU is the bench's alone, profiles/r06_bench_default.json.  Under this suite other processes' kernels
    # share the CUs, and a workgroup of the five-class kernel - 112 KB of LDS - cannot sit beside a 64 KB block of theirs where
    # the one-class kernel's 91 KB can: readings of 4 - 6x were seen here, so only the order of magnitude is held)
    assert 0.9 * cf["cfg5"]["ms_per_step"] < cf["cfg5_c5"]["ms_per_step"] < 20 * cf["cfg5"]["ms_per_step"]
    # what one GPU's numbers say about eight: the strong-scaling ceilings, stated by the line itself
    sb = d["strong_bound"]
    assert abs(sb["cfg3_65536_over_8"] - cf["cfg3_b65536"]["ms_per_step"] / cf["cfg3"]["ms_per_step"]) < 0.02
    assert abs(sb["cfg5_256_restarts_over_8"] - cf["cfg5"]["ms_per_step"] / cf["cfg5_shard32"]["ms_per_step"]) < 0.02
    assert 1.0 < sb["cfg3_65536_over_8"] < 8.0 and 0.0 < sb["cfg5_256_restarts_over_8"] < 8.0   # (the shard's cooperative launch is the
    #                                  reading that other processes on the GPU disturb, see below: its bound is held when it is retaken)
    # and the other end of the settled `value`: the first launches after the GPU sat idle
    cold = d["callers"]["headline_cold_us"]
    assert "error" not in cold, cold
    assert len(cold["first"]) == 3 and cold["settled"] > 0 and cold["mean_first"] > 0.9 * cold["settled"]
    # the 8-GPU shard of config #5 runs its paths on several workgroups each (cluster form): well under the 256-restart time
    # (12.0 against 28.3 us).  The cluster form is a COOPERATIVE launch: it waits for the whole GPU, so once - 1 of 5 suite runs -
    # it read 67 us behind a previous test's processes that were still winding down; such a reading is taken again, once
    if not cf["cfg5_shard32"]["ms_per_step"] < 0.7 * cf["cfg5"]["ms_per_step"]:
        d = _run(["--no-cpu-baseline"], steps=10, warmup=3, timeout=900)
        cf = d["configs"]
    assert cf["cfg5_shard32"]["ms_per_step"] < 0.7 * cf["cfg5"]["ms_per_step"]
    assert d["settle_steps"] > 0
    # the caller beside the path: fit_poly's solve in one launch, a solution (residual of K x = y) and not slower than the library
    fs = d["callers"]["fit_poly_solve"]
    for key in ("S438", "S2000"):
        assert 0 < fs[key]["dcx_solve"] < 1.5 * fs[key]["hipsolver"], fs   # (measured 0.5x / 0.7x; a loop of 10 on a busy box is noisy)
        assert fs[key]["dcx_solve_residual"] <= max(1e-4, fs[key]["hipsolver_residual"]), fs   # (fp64 inside: the smaller one)
    # the Python boundary: poly_score without a gradient costs what the raw call costs (host-side: one launch either way), and
    # with autograd behind it not more than torch's own engine floor + the forward + a margin for this package's Function
    ps = d["callers"]["poly_score_us"]
    assert "error" not in ps, ps
    for B in ("B20", "B50", "B256", "B4096"):
        assert 0 < ps[B]["raw"] and ps[B]["fwd"] < ps[B]["raw"] + 25.0, ps          # (measured: + 0 ... 3 us)
        assert ps[B]["score_and_grad"] < ps[B]["raw"] + 25.0, ps                     # (measured: + 0 ... 3 us)
        # (through torch's autograd engine: 70 - 170 us on the pool's hosts where torch alone needs 60 - 80; bounded loosely)
        assert ps[B]["fwd_bwd"] < 4.0 * ps["torch_autograd_floor"] + 50.0, ps
    # the escape loop as one library call: both routes ran, all three steps were taken, and the fused form is the faster one
    es = d["callers"]["escape_us"]
    assert "error" not in es, es
    assert es["routes"] == ["fused", "host"] and es["evaluations"] == 3 and 0 < es["fused"] < es["host_loop"], es
    assert es["batch_65536x20"]["M_escapes_per_s"] > 5.0, es
    rf = d["roofline"]
    ck = rf["clock"]   # measured inside the kernel beside the loop's launches: the part does not hold 2.4 GHz under this load
    assert "error" not in ck, ck
    assert 0.8 <= ck["shader_ghz_under_this_load"] <= ck["shader_ghz_idle"] + 0.05 <= 2.6, ck
    assert rf["frac_at_measured_clock"] >= rf["frac"] * 0.99
    for name, c in cf.items():
        assert "error" not in c, (name, c)
        assert c["value"] > 0 and 0 < c["frac"] < 1 and c["kernel_ms"] <= c["ms_per_step"] * 1.05, (name, c)


def test_cpu_baseline_leg():
    d = _run(["--batch", "4096"], steps=5, warmup=2)
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["value"] > 0


@pytest.mark.parametrize("extra", [["--scaling", "weak"], ["--scaling", "strong"], ["--workload", "cfg3", "--scaling", "strong"],
                                   ["--workload", "cfg5", "--scaling", "strong"], ["--workload", "cfg5", "--scaling", "weak"]],
                         ids=["headline-weak", "headline-strong", "cfg3-strong", "cfg5-strong", "cfg5-weak"])
def test_two_rank_rehearsal_on_one_gpu(extra):
    """world size 2 through torch.distributed.run exactly as the driver launches it, both ranks on cuda:0 with the gloo
    backend (DCX_BENCH_SAME_GPU=1: RCCL refuses two ranks on one device, so the gather goes through host buffers).
    Timings mean nothing here; the shard arithmetic, the barriers and reductions, the variants and the one JSON line of
    rank 0 are the code the 2/4/8-GPU runs execute."""
    env = dict(os.environ, DCX_BENCH_SAME_GPU="1")
    port = str(29600 + (abs(hash(tuple(extra))) % 200))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12",
                        "--warmup", "2"] + extra, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert (KEYS - {"cpu_baseline"}) <= set(d) and d["n_gpus"] == 2 and d["steps"] == 12 and d["value"] > 0
    assert d["multi"]["ranks"] == 2 and d["scaling"] == extra[extra.index("--scaling") + 1]
    cfg = d["config"]
    if "cfg5" in extra:
        assert d["multi"]["gather"] == "summaries"
        assert cfg["global_batch"] == (256 * 50 if d["scaling"] == "strong" else 2 * 256 * 50)
    elif d["scaling"] == "strong":
        assert cfg["global_batch"] == 65536 and cfg["batch_per_gpu"] == 32768
    else:
        assert cfg["global_batch"] == 2 * cfg["batch_per_gpu"] == 2 * 65536
    assert d["variants"]["other_scaling"]["value"] > 0


def test_bare_command_with_gpus_2_launches_its_own_ranks():
    """`python3 bench.py --gpus 2 --steps 20 --warmup 5` with NO launcher around it and no WORLD_SIZE in
    the environment re-executes itself under torch.distributed.run and prints the one line of a two-rank job"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["DCX_BENCH_SAME_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["value"] > 0
    assert d["multi"]["ranks"] == 2 and d["multi"]["primary"]["gather"] == "per-call"
    assert d["config"]["global_batch"] == 2 * 65536 and d["scaling"] == "weak"
    assert d["cpu_baseline"] is None   # (the CPU baseline is an N = 1 measurement)


def _one_line(stdout):
    lines = [ln for ln in stdout.split("\n") if ln.strip().startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_the_line_survives_an_abort_after_the_timed_region():
    """fault injection: the process is aborted - as a library thread would, no Python clean-up - right
    after the timed region, before any variant: the keeper process still prints the primary line, exactly once"""
    env = dict(os.environ, MASTER_PORT="29549", DCX_BENCH_FAULT="after_primary")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--force-dist", "--batch", "4096"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0
    d = _one_line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["multi"]["ranks"] == 1 and d["value"] > 0
    assert d["cpu_baseline"]["value"] > 0 and "variants" not in d


def test_two_rank_line_survives_an_abort_inside_the_graph_capture():
    """the same with two ranks launched as the driver launches them, the abort INSIDE the construction of the captured
    gather - the last variant: one valid line with the primary numbers (n_gpus, multi.ranks), no variants"""
    env = dict(os.environ, DCX_BENCH_SAME_GPU="1", DCX_BENCH_FAULT="capture")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29877", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12",
                        "--warmup", "2", "--batch", "8192"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode != 0
    d = _one_line(r.stdout)
    assert (KEYS - {"cpu_baseline"}) <= set(d) and d["n_gpus"] == 2 and d["multi"]["ranks"] == 2 and d["value"] > 0
    assert d["multi"]["gather"] == "per-call" and "variants" not in d


def test_the_line_survives_a_hang_in_the_side_measurements():
    """round 5: a side measurement that never returns (injected: every rank sleeps after its primary numbers are safe) does not
    run into the driver's timeout - each rank's timer ends it with exit code 0 and the keeper prints the primary line"""
    env = dict(os.environ, DCX_BENCH_SAME_GPU="1", DCX_BENCH_FAULT="hang", DCX_BENCH_SIDE_BUDGET_S="5")
    env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "2", "--batch", "8192"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 2 and d["multi"]["ranks"] == 2 and d["value"] > 0 and "variants" not in d
