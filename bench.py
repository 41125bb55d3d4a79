#!/usr/bin/env python3
"""bench.py — throughput of the fused score+grad hot path on MI355X.

Metric (BASELINE.json): million collision-score+grad evaluations per second, 7-DoF FK kernel, 2k supports.
One evaluation = one configuration q[7] -> score[C] and d(sum_c upstream*score)/dq[7] against all S supports.
A "step" = one pass of the hot path (ONE `dcx_score_grad` launch; config #5: one fused Adam iteration) over one batch
of synthetic configurations already resident in HBM.

    python bench.py                      # 1 GPU, headline workload
    python bench.py --gpus N --steps K --warmup W    # N > 1 without a launcher: becomes its own (self_launch) ...
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--gather per-call|overlapped|bucketed|none]   # ... or under one

N > 1: one process per GPU, the model replicated, the configuration batch sharded, no data-path collective; the scores
of EVERY call are all-gathered over RCCL/xGMI (`--gather per-call`, the default: in order on the launch stream, what a
consumer that needs the scores before its next call sees; `overlapped`: the gather of call i runs beside the
sweep of call i+1 on the process group's stream; `graph`: the same inside a captured HIP graph).  `--scaling weak` (default) keeps the per-GPU batch fixed as N grows,
`--scaling strong` divides the workload's fixed global batch (65536 for headline / config #3, 256 restarts for
config #5) by N.  With N > 1 the line also carries short measurements of the other variants (`variants`): the
other scaling mode, the overlapped gather, the gather bucketed every 4 calls, and no gather.  The per-call form is measured and
parked first (`multi.primary`); of the forms that deliver every call's scores (per-call, overlapped, graph), each timed over
exactly K steps, the fastest that completed is the line's `value` (`multi.gather`).  Side measurements that hang end after
DCX_BENCH_SIDE_BUDGET_S (240 s) with the line measured so far.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` and `cpu_baseline`.  The line cannot
be lost to a side measurement: rank 0 hands the PRIMARY line (timed region + CPU baseline) to a keeper process as soon as
it exists, and the complete line (with `variants` / `configs`) at the end; the keeper prints the complete line if it
arrived and the primary one if this process died first (`LineKeeper`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_TFLOPS = 157.3   # MI355X fp32 vector peak == fp32 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (robot, kernel (kind,p0,p1), S, C, per-GPU batch (weak), global batch (strong), description)
    "headline": ("baxter", (1, 1.0, 1.0), 2000, 1, 65536, 65536,
                 "7-DoF Baxter DH chain (D=12), Polyharmonic(1,1), S=2000, C=1 (SURVEY.md §8d headline)"),
    "headline_rq": ("baxter", (0, 10.0, 2.0), 2000, 1, 65536, 65536,
                    "DiffCo.score at the metric's size: RQKernel(10) gains, 7-DoF Baxter (D=12), S=2000, C=1 (kernel_perceptrons.py:359-370)"),
    "cfg2": ("baxter", (1, 1.0, 1.0), 1000, 1, 4096, 4096, "BASELINE config #2: 7-DoF, FK kernel, 1k supports, batch 4096"),
    "cfg2_panda": ("panda", (1, 1.0, 1.0), 1000, 1, 4096, 4096, "config #2 with PandaFK (D=21)"),
    "cfg3": ("baxter", (0, 10.0, 2.0), 2000, 5, 8192, 65536,
             "BASELINE config #3: MultiDiffCo C=5, RQ(10), S=2000, batch 65536 over 8 GPUs = 8192 per GPU"),
    "cfg3_poly": ("baxter", (1, 1.0, 1.0), 2000, 5, 8192, 65536,
                  "config #3 with the Polyharmonic(1,1) spline nodes MultiDiffCo.rbf_score evaluates "
                  "(deprecated/MultiDiffCo.py:156-169): C=5, S=2000, 8192 per GPU"),
    "cfg3_c8": ("baxter", (0, 10.0, 2.0), 2000, 8, 8192, 65536,
                "config #3 with eight classes (the widest compiled class count; matrix-core A/B, profiles/r03_mfma_ab.txt)"),
    "cfg4": (None, (0, 10.0, 2.0), 10000, 1, 1 << 20, 1 << 20, "BASELINE config #4: SE(3) no-FK (D=6), RQ(10), S=10k, 1M configs"),
    # config #5: a "step" is ONE fused Adam iteration over R restarts x 50 waypoints (= 50 R score+grad evals)
    "cfg5": ("baxter", (1, 1.0, 1.0), 2000, 1, 256 * 50, 256 * 50,
             "BASELINE config #5: fused Adam trajopt, 7-DoF, 50 waypoints x 256 restarts, S=2000 "
             "(step = 1 iteration: score+hinge-grad sweep + Adam step; one persistent launch per <= 192 iterations)"),
    # the same loop on config #3's model: five classes under per-class margins (optim.py:88-89 with a [C] safety margin, as
    # scripts/2d_trajopt.py:94-102 / scripts/active.py:28-121 run it on a MultiDiffCo): two sweeps per iteration (class scores, then
    # the gradient whose upstream is the hinge's indicator), dcx_traj_adam_run_mc
    "cfg5_c5": ("baxter", (1, 1.0, 1.0), 2000, 5, 256 * 50, 256 * 50,
                "config #5's Adam loop on config #3's model: MultiDiffCo C=5 Polyharmonic nodes, S=2000, 50 waypoints x 256 restarts, "
                "collision term sum_c clamp(score_c - margin_c, 0) (step = 1 iteration = two sweeps + Adam step)"),
}
TRAJ_W = 50
GATHER_EVERY = 4  # --gather bucketed: calls per all-gather
PROMOTABLE = ("per-call", "overlapped", "graph")  # forms that deliver the gathered scores of EVERY call (see "Promotion" in main)
SETTLE_STEPS = 24  # first untimed launches before the W warm-up steps: one-off costs, and the estimate for the settle phase
# untimed launches keep the GPU busy this long before the warm-up (clock / power-state ramp, see measure()).  Round 5: 100 -> 250 ms:
# the FIRST bench process on a lease (the GPU asleep for the 10 - 25 s of the CPU baseline before it) read 2 - 12 % slower than
# the runs after it with 100 ms (profiles/r05_bench_driver_cmd.jsonl, first line); developer override DCX_BENCH_SETTLE_MS
SETTLE_MS = float(os.environ.get("DCX_BENCH_SETTLE_MS", "250"))


SAME_GPU = os.environ.get("DCX_BENCH_SAME_GPU", "") not in ("", "0")
"""Developer rehearsal of the N > 1 code path on a box with ONE GPU: every rank uses cuda:0 and the gloo backend (RCCL
refuses two ranks on one device), the score gather goes through host buffers.  Timings are meaningless; shard
arithmetic, the variants, the reductions and the JSON assembly are the real ones (tests/test_gpu_bench_contract.py)."""


class _Done:
    def wait(self):
        return True


_KEEPER_SRC = r"""
import signal, sys
for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
    signal.signal(sg, signal.SIG_IGN)   # the launcher tears the job down after a rank died: the line still goes out
best = None
for ln in sys.stdin:
    ln = ln.rstrip("\n")
    if ln.startswith("P ") and best is None:
        best = ln[2:]
    elif ln.startswith("F "):
        best = ln[2:]
if best is not None:
    sys.stdout.write(best + "\n")
    sys.stdout.flush()
"""


class LineKeeper:
    """Owner of the ONE line on the real stdout.  A tiny child process (plain Python, no torch, no GPU) reads this
    process's pipe: `primary(line)` parks the line of the timed region, `final(line)` the complete one; when the pipe closes
    - normally, or because this process was aborted from a library thread in the middle of a side measurement (c10d's
    watchdog does that when a graph capture of a collective goes wrong; no Python handler runs then) - the keeper prints
    the complete line if it has one, else the primary line.  Either way exactly one line, and never none once the timed
    region is over."""

    def __init__(self, fd):
        import subprocess
        self.proc = subprocess.Popen([sys.executable, "-c", _KEEPER_SRC], stdin=subprocess.PIPE, stdout=fd, close_fds=True)

    def _send(self, tag, obj):
        self.proc.stdin.write((tag + " " + json.dumps(obj) + "\n").encode())
        self.proc.stdin.flush()

    def primary(self, obj):
        self._send("P", obj)

    def final(self, obj):
        self._send("F", obj)

    def close(self):
        try:
            self.proc.stdin.close()
            self.proc.wait(timeout=30)
        except Exception:   # noqa: BLE001  (the line is the keeper's business from here on; never turn it into a failed run)
            pass


def _fault(where):
    """developer fault injection (tests/test_gpu_bench_contract.py): DCX_BENCH_FAULT=<where> kills the process the way a
    library thread would (abort(), no Python clean-up)"""
    if os.environ.get("DCX_BENCH_FAULT", "") == where:
        sys.stderr.write(f"bench: injected fault at {where}\n")
        sys.stderr.flush()
        os.abort()


def gather_scores(full, local, async_op=False):
    if not SAME_GPU:
        return dist.all_gather_into_tensor(full, local, async_op=async_op)
    torch.cuda.current_stream(local.device).synchronize()
    host = torch.empty(full.shape, dtype=full.dtype)
    dist.all_gather_into_tensor(host, local.cpu())
    full.copy_(host)
    return _Done() if async_op else None


def flops_per_eval(D, C, S):
    """F_pair = 5D + 4C + 6, F_eval = S*F_pair + 800 (FK + J^T)"""
    return S * (5 * D + 4 * C + 6) + 800


def bytes_per_eval(dof, C):
    """SURVEY.md §8d algorithmic HBM bytes: q in, score out, grad out"""
    return 4 * dof + 4 * C + 4 * dof


def make_workload(name, batch, dev, seed=0):
    from diffco_amd import _fkdesc, _ops, model
    rob_name, kspec, S, C, B, _, desc_txt = WORKLOADS[name]
    B = batch or B
    g = torch.Generator().manual_seed(0)               # the model (supports, weights) is the same on every rank
    gq = torch.Generator().manual_seed(1000 + seed)    # the batch shard is the rank's own
    rob = None
    if rob_name is None:
        lo = torch.tensor([-10.0] * 3 + [-np.pi] * 3)
        hi = -lo
        desc = _fkdesc.none_desc(6)
    else:
        rob = {"baxter": model.BaxterLeftArmFK, "panda": model.PandaFK}[rob_name]()
        lo, hi = rob.limits[:, 0], rob.limits[:, 1]
        desc = rob.fk_desc()
    sup_q = torch.rand((S, len(lo)), generator=g) * (hi - lo) + lo
    W = torch.randn((S, C), generator=g)
    if name in ("cfg3", "cfg3_poly", "cfg3_c8", "cfg5_c5"):  # 40 % of the entries zeroed per class (mimics deprecated/MultiDiffCo.py:152-153)
        W = W * (torch.rand((S, C), generator=g) >= 0.4)
    q = torch.rand((B, len(lo)), generator=gq) * (hi - lo) + lo
    sup = _ops.fkine(desc, sup_q.to(dev)).reshape(S, -1)
    m = _ops.ScoreModel(desc, *kspec, sup, W.to(dev), device=dev)
    return dict(name=name, text=desc_txt, model=m, desc=desc, kspec=kspec, S=S, C=C, B=B, D=desc.feature_dim,
                dof=desc.dof, q=q.to(dev).contiguous(), sup=sup, W=W, q_cpu=q, rob_name=rob_name, rob=rob, lo=lo, hi=hi)


def traj_state(w, R, Wp, dev):
    """config #5 trajectory state for direct `dcx_traj_adam_run` calls (developer tools): R restarts of Wp waypoints in
    joint limits; grad_tol = 0 so no path ever freezes"""
    import ctypes as Ct
    from diffco_amd import _lib
    dof = w["dof"]
    f32 = dict(device=dev, dtype=torch.float32)
    path = w["q"][:R * Wp].reshape(R, Wp, dof).clone()
    bufs = [path, torch.zeros_like(path), torch.zeros_like(path),
            torch.stack([w["lo"], w["hi"]], dim=1).to(**f32).contiguous(), torch.empty(R * Wp, **f32),
            torch.empty((R * Wp, dof), **f32), torch.zeros((R, 8), **f32), torch.full((R,), float("inf"), **f32),
            torch.full((R,), float("inf"), **f32), path.clone(), torch.full((R,), float("inf"), **f32), path.clone(),
            torch.zeros(R, device=dev, dtype=torch.int32), torch.zeros(R, device=dev, dtype=torch.int32)]
    tst = _lib.TrajState(R, Wp, *(Ct.c_void_p(t.data_ptr()) for t in bufs))
    topt = _lib.TrajOpts(0.05, 0.9, 0.999, 1e-8, 1, 10, 10, 10, 0.0, 0.3, 1e-2, 0.0)
    return tst, topt, bufs


def cpu_baseline(w, budget_s=12.0):
    """The CPU oracle (oracle/, a C/OpenMP port of the reference algorithm) timed on this box's host
    cores on a bounded sample of the same workload.  Reported beside the GPU number, never the target."""
    from oracle import oracle
    threads = oracle.max_threads()
    sup = w["sup"].cpu().numpy()
    Wn = w["W"].numpy()
    k = w["kspec"]
    n0 = 2048
    t0 = time.perf_counter()
    oracle.score_grad(w["desc"], *k, sup, Wn, w["q_cpu"][:n0].numpy())  # warm-up + rate estimate
    dt = time.perf_counter() - t0
    # one timed pass of ~budget/3 seconds (at least the warm-up size); three passes, best kept
    n = int(max(n0, n0 * (budget_s / 3.0) / max(dt, 1e-6)))
    reps_q = -(-n // len(w["q_cpu"]))
    qs = (w["q_cpu"].repeat(reps_q, 1)[:n] if reps_q > 1 else w["q_cpu"][:n]).numpy()
    best, reps = 1e30, 0
    for _ in range(3):
        t0 = time.perf_counter()
        oracle.score_grad(w["desc"], *k, sup, Wn, qs)
        best = min(best, time.perf_counter() - t0)
        reps += 1
    return {"value": round(n / best / 1e6, 5), "unit": "M evals/s", "cores": threads, "kind": "port",
            "sample": f"{n} configs of the same workload (S={w['S']}, D={w['D']}, C={w['C']}), fp32 C/OpenMP oracle, "
                      f"best of {reps}", "seconds": round(best, 3)}


def _torch_dh_fkine(desc):
    """differentiable torch-CPU forward kinematics of a single DH chain from its description: the reference's expression
    (DH2mat -> cumulative products -> translation columns of the masked frames) as torch ops"""
    n = desc.chain_len[0]
    col = lambda arr: torch.tensor([arr[0][i] for i in range(n)], dtype=torch.float32)
    a_, d_, sa_, ca_, t0_ = col(desc.a), col(desc.d), col(desc.sin_alpha), col(desc.cos_alpha), col(desc.theta0)
    frames = [desc.pt_frame[k] for k in range(desc.n_points)]
    offs = [[desc.pt_off[k][j] for j in range(3)] + [1.0] for k in range(desc.n_points)]

    def fkine(q):
        th = q + t0_
        c, s_ = th.cos(), th.sin()
        z, o = torch.zeros_like(c), torch.ones_like(c)
        A = torch.stack([torch.stack([c, -s_ * ca_, s_ * sa_, a_ * c], -1), torch.stack([s_, c * ca_, -c * sa_, a_ * s_], -1),
                         torch.stack([z, sa_ * o, ca_ * o, d_ * o], -1), torch.stack([z, z, z, o], -1)], 2)  # [B, n, 4, 4]
        T, cum = A[:, 0], [A[:, 0]]
        for i in range(1, n):
            T = torch.bmm(T, A[:, i])
            cum.append(T)
        return torch.stack([(cum[f] @ torch.tensor(off))[:, :3] for f, off in zip(frames, offs)], 1)
    return fkine


def _torch_cpu_leg(path, nthr, n, budget):
    """child process of torch_cpu_baseline: one timing of the reference expression with `nthr` torch threads, nothing else
    alive in the process (no OpenMP team of the oracle, no GPU context).  Prints one JSON object."""
    torch.set_num_threads(nthr)
    z = np.load(path)
    sup, Wt, q_all = torch.from_numpy(z["sup"]), torch.from_numpy(z["W"]), torch.from_numpy(z["q"])
    kspec = [float(v) for v in z["kspec"]]
    rob_name = str(z["rob"])
    fk = None
    if rob_name != "none":
        from diffco_amd import model
        fk = _torch_dh_fkine({"baxter": model.BaxterLeftArmFK, "panda": model.PandaFK}[rob_name]().fk_desc())
    q0 = q_all[:n].clone()

    def run():
        q = q0.clone().requires_grad_(True)
        x = q if fk is None else fk(q).reshape(n, -1)
        if int(kspec[0]) == 0:
            kv = 1 / (1 + kspec[1] / kspec[2] * torch.cdist(x, sup).square()) ** kspec[2]
        else:
            kv = torch.cdist(x, sup) / kspec[2]
        (kv @ Wt).sum().backward()
        return q.grad
    t0 = time.perf_counter()
    run()
    best, t_start, reps = time.perf_counter() - t0, time.perf_counter(), 0
    while reps < 5 and time.perf_counter() - t_start < budget:
        t0 = time.perf_counter()
        run()
        best = min(best, time.perf_counter() - t0)
        reps += 1
    print(json.dumps({"value": round(n / best / 1e6, 5), "cores": nthr, "configs": n, "reps": reps + 1}))


def torch_cpu_baseline(w, budget_s=6.0):
    """A torch-CPU restatement of the reference EXPRESSION on this box's host cores, as SURVEY.md §8d specifies it: FK ->
    cdist -> kernel -> matmul, `.sum().backward()` down to the joint angles, `torch.set_num_threads(os.cpu_count())`.
    torch's intra-op pool does not scale to a 256-thread host at this size, so the expression is timed twice: with every core
    as specified (`all_cores`) and with 32 threads; `value` is the better of the two.  Each timing runs in a FRESH process
    (OMP_NUM_THREADS = its thread count): inside this one the oracle's OpenMP team is already spun up on every core, and 256
    torch threads beside it measured oversubscription, not the expression (0.00016 vs 0.29 M evals/s).
    Secondary information beside `cpu_baseline` (the C/OpenMP oracle)."""
    import subprocess
    import tempfile
    if w["kspec"][0] not in (0, 1):
        return None
    ncpu = os.cpu_count() or 1
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "leg.npz")
        np.savez(path, sup=w["sup"].cpu().numpy(), W=w["W"].numpy(), q=w["q_cpu"][:4096].numpy(),
                 kspec=np.array(w["kspec"], dtype=np.float64), rob=np.array(w["rob_name"] or "none"))

        def timed(nthr, n, budget):
            env = dict(os.environ, OMP_NUM_THREADS=str(nthr), MKL_NUM_THREADS=str(nthr), HIP_VISIBLE_DEVICES="")
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--torch-cpu-leg", path, str(nthr), str(n), str(budget)],
                                   capture_output=True, text=True, timeout=120 + 20 * budget, env=env, cwd=ROOT)
                return json.loads([ln for ln in r.stdout.split("\n") if ln.startswith("{")][-1])
            except Exception as exc:  # noqa: BLE001  (a side measurement)
                return {"value": 0.0, "cores": nthr, "configs": n, "reps": 0, "error": f"{type(exc).__name__}: {exc}"[:160]}
        capped = timed(min(ncpu, 32), min(4096, w["B"]), budget_s / 2)
        allc = timed(ncpu, min(1024, w["B"]), budget_s / 2) if ncpu > 32 else capped
    best = max(capped, allc, key=lambda r: r["value"])
    if best["value"] <= 0:
        return None
    return {"value": best["value"], "unit": "M evals/s", "cores": best["cores"], "all_cores": allc, "capped_32": capped,
            "sample": f"torch {torch.__version__} CPU, one fresh process per timing: " + ("FK -> " if w["rob_name"] is not None else "") +
                      f"cdist -> kernel -> matmul -> backward to q, {best['configs']} configs, best of {best['reps']}"}


def load_profile_json(fname):
    p = os.path.join(ROOT, "profiles", fname)
    try:
        with open(p) as f:
            return json.load(f)
    except Exception:
        return None


# ------------------------------------------------------------------------------------------------- measured loops
class ScoreLoop:
    """K calls of `dcx_score_grad` on this rank's batch shard, with the requested all-gather policy for the scores"""

    def __init__(self, w, dev, world, gather):
        import ctypes as Ct
        from diffco_amd import _lib
        self.Ct, self._lib, self.lib = Ct, _lib, _lib.load()
        self.w, self.dev, self.world, self.gather = w, dev, world, gather
        B, C, dof = w["B"], w["C"], w["dof"]
        self.B = B
        f32 = dict(device=dev, dtype=torch.float32)
        self.grad = torch.empty((B, dof), **f32)
        self.qp, self.gp = Ct.c_void_p(w["q"].data_ptr()), Ct.c_void_p(self.grad.data_ptr())
        self.K = GATHER_EVERY if gather == "bucketed" else 1
        nbuf = 2 if gather in ("overlapped", "bucketed") else 1
        self.local = [torch.empty((self.K * B, C), **f32) for _ in range(nbuf)]
        self.full = [torch.empty((world * self.K * B, C), **f32) for _ in range(nbuf)] if gather != "none" else None
        self.overlap = gather in ("overlapped", "bucketed")
        self.pending = [None] * nbuf
        self._gather_ms = None
        self.graph, self.G = None, 8
        if gather == "graph":
            if SAME_GPU:   # the rehearsal's host-buffer gather cannot be captured
                _fault("capture")
                self.gather = "per-call"
            else:
                self._capture()
                # every rank takes the same form: one rank replaying graphs against peers issuing per-call gathers would
                # still match collective for collective, but the timings would describe neither
                if dist.is_available() and dist.is_initialized():
                    ok = torch.tensor([1 if self.graph is not None else 0], device=dev, dtype=torch.int32)
                    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                    if int(ok.item()) == 0 and self.graph is not None:
                        print("bench: a peer could not capture the gather; per-call gather on every rank", file=sys.stderr)
                        self.graph, self.gather = None, "per-call"

    def _launch(self, out):
        Ct, w = self.Ct, self.w
        st = Ct.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        self._lib.check(self.lib.dcx_score_grad(w["model"]._h, self.qp, self.B, None, Ct.c_void_p(out.data_ptr()), self.gp, st))

    def _capture(self):
        """G steps - sweep i on the launch stream, the all-gather of its scores on a side stream behind an event, sweep i + 1
        not waiting for it - captured ONCE into a HIP graph: a replay costs the host one call per G steps, where the eager
        overlapped form pays c10d's bookkeeping (27 us) on every call.  The gather of step i only has to be over before
        step i + 2 rewrites its buffer.  Falls back to the per-call form if the collective cannot be captured here."""
        dev = self.dev
        _fault("capture")
        self.local = [torch.empty((self.B, self.w["C"]), device=dev, dtype=torch.float32) for _ in range(2)]
        self.full = [torch.empty((self.world * self.B, self.w["C"]), device=dev, dtype=torch.float32) for _ in range(2)]
        main, side = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        try:
            with torch.cuda.stream(main):   # warm-up on the capture stream: the model's split scratch for it, RCCL's buffers
                for b in (0, 1, 0, 1):
                    self._launch(self.local[b])
                    gather_scores(self.full[b], self.local[b])
            torch.cuda.synchronize(dev)
            # c10d's watchdog thread polls the events of earlier collectives (the warm-up's, an eager variant's): under the
            # default global capture mode such a query from ANOTHER thread invalidates the capture and takes the process
            # down from the watchdog (hipErrorStreamCaptureUnsupported).  Thread-local mode confines the capture's rules to
            # this thread; the short sleep lets the watchdog retire what is already complete.
            time.sleep(0.3)
            g = torch.cuda.CUDAGraph()
            done = [None, None]
            with torch.cuda.graph(g, stream=main, capture_error_mode="thread_local"):
                for i in range(self.G):
                    b = i % 2
                    if done[b] is not None:
                        main.wait_event(done[b])
                    self._launch(self.local[b])
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        gather_scores(self.full[b], self.local[b])
                        done[b] = torch.cuda.Event()
                        done[b].record(side)
                main.wait_event(done[0])
                main.wait_event(done[1])
            torch.cuda.synchronize(dev)
            g.replay()
            torch.cuda.synchronize(dev)
            self.graph = g
        except Exception as exc:  # noqa: BLE001  (no graph: the in-order per-call gather, which always works)
            print(f"bench: graph capture of sweep + all-gather failed ({type(exc).__name__}: {str(exc)[:120]}); per-call gather", file=sys.stderr)
            torch.cuda.synchronize(dev)
            self.graph, self.gather = None, "per-call"

    def step(self, i, last):
        Ct, w = self.Ct, self.w
        K, B = self.K, self.B
        if self.gather == "graph":  # a stray step outside whole replays: in order
            self._launch(self.local[0])
            gather_scores(self.full[0], self.local[0])
            return
        b = (i // K) % len(self.local)
        if self.overlap and i % K == 0 and self.pending[b] is not None:
            self.pending[b].wait()  # this buffer's previous gather must be done before the sweep rewrites it (a stream wait)
            self.pending[b] = None
        out = self.local[b][(i % K) * B:(i % K + 1) * B]
        st = Ct.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        self._lib.check(self.lib.dcx_score_grad(w["model"]._h, self.qp, B, None, Ct.c_void_p(out.data_ptr()), self.gp, st))
        if self.gather == "none" or not (i % K == K - 1 or i == last):
            return
        if self.gather == "per-call":
            # the consumer needs the gathered scores before it issues the next call: the launch stream waits for the collective
            gather_scores(self.full[0], self.local[0])
            return
        # overlapped: the process group runs the collective on its own stream behind an event of the launch stream (what
        # ProcessGroupNCCL does for async_op=True); the launch stream only waits for it when the buffer comes round again.
        # No side stream, context switch or timing events of our own per call: at ~100 us per sweep every host microsecond of
        # the step shows (the r02 version with its own side stream and three events per call cost 24 % on one rank).
        self.pending[b] = gather_scores(self.full[b], self.local[b], async_op=True)

    def drain(self):
        for k, h in enumerate(self.pending):
            if h is not None:
                h.wait()
                self.pending[k] = None

    def run(self, n, timed=False):
        if self.graph is not None:
            for _ in range(n // self.G):
                self.graph.replay()
            n = n % self.G
        for i in range(n):
            self.step(i, n - 1)

    def gather_ms(self):
        """the collective alone (launch-stream HIP events around blocking gathers, outside the timed region)"""
        if self.gather == "none" or self.full is None:
            return None
        if self._gather_ms is None:
            self.drain()
            torch.cuda.synchronize(self.dev)
            gather_scores(self.full[0], self.local[0])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                gather_scores(self.full[0], self.local[0])
            e1.record()
            torch.cuda.synchronize(self.dev)
            self._gather_ms = e0.elapsed_time(e1) / n
        return self._gather_ms


class TrajLoop:
    """config #5 through the library path: restarts sharded over the ranks by `diffco_amd.traj.ShardedAdamRun`, K
    iterations enqueued natively, only the per-restart summaries (and candidate paths) gathered at the end"""

    def __init__(self, w, dev, world, n_total, group):
        from diffco_amd.traj import ShardedAdamRun
        dof = w["dof"]
        g = torch.Generator().manual_seed(4242)  # the SAME restarts on every rank; each rank takes its slice
        lo, hi = w["lo"], w["hi"]
        inits = torch.rand((n_total, TRAJ_W, dof), generator=g) * (hi - lo) + lo
        self.run_obj = ShardedAdamRun(w["model"], torch.stack([lo, hi], dim=1), inits, lr=0.05, safety_margin=0.0,
                                      max_speed=0.3, grad_tol=0.0, group=group, sharded=world > 1)
        self.B = self.run_obj.R * TRAJ_W
        self.gather = "summaries"

    def run(self, n, timed=False):
        self.run_obj.run(n)
        # the job's only exchange: 4 floats + two candidate paths per restart.  It closes the timed region (whole-job
        # time) and is also issued once after the warm-up iterations, so that the torch kernels / collectives it uses are
        # loaded before the clock starts.
        self.summary = self.run_obj.finish()

    def drain(self):
        pass

    def gather_ms(self):
        return None


def hwmon_read(dev_index=0):
    """(sclk MHz, socket power W) from amdgpu's hwmon files, None where sysfs has nothing.  What freq1_input means differs between
    the leases of this pool (a DPM level on some, an average on others: 2400, 2010 and 1410 MHz have all been read during the
    same 85 - 87 us launches), so the line's clock is the in-kernel probe's; these two numbers ride along as `hwmon`."""
    import glob
    pick = lambda pat: (sorted(glob.glob(pat)) or [None])[0]
    base = f"/sys/class/drm/card{dev_index}/device/hwmon/hwmon*/"

    def rd(path, scale):
        try:
            with open(path) as f:
                return float(f.read().strip()) / scale
        except Exception:   # noqa: BLE001
            return None
    f_clk, f_pow = pick(base + "freq1_input"), pick(base + "power1_input") or pick(base + "power1_average")
    return (rd(f_clk, 1e6) if f_clk else None), (rd(f_pow, 1e6) if f_pow else None)


def clock_under_load(loop, dev, ms=6.0):
    """The clock the shaders have WHILE the measured loop runs: `dcx_debug_clock_probe` - one wave that samples s_memtime (the
    shader clock counter) and s_memrealtime (a fixed 100 MHz) at its start and `ms` later - goes to a side stream, the loop's
    launches keep the launch stream busy meanwhile; (shader ticks) / (wall ticks) x rate = GHz.  Run AFTER the timed region
    (the probe occupies one wave slot of one CU).  Also the same probe on an idle GPU, for reference."""
    import ctypes as Ct
    from diffco_amd import _lib
    lib = _lib.load()
    out = torch.zeros(4, device=dev, dtype=torch.int64)
    khz = Ct.c_int32(0)
    side = torch.cuda.Stream(dev)

    def probe(busy):
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 0
        if busy:
            loop.run(SETTLE_STEPS)      # (the probe starts into a GPU that is already under the load)
        _lib.check(lib.dcx_debug_clock_probe(dev.index, Ct.c_void_p(out.data_ptr()), int(ms * 1e-3 * 100e6), Ct.byref(khz),
                                             Ct.c_void_p(side.cuda_stream)))
        if busy:
            e0.record()
            loop.run(SETTLE_STEPS)
            e1.record()
            e1.synchronize()
            per = max(e0.elapsed_time(e1) / SETTLE_STEPS, 1e-3)
            n = int(min(max(1.3 * ms / per, 1), 20000))
            loop.run(n)
            # (the launches are queued: the GPU works them off while the host looks at the driver's sensors)
            for _ in range(3):
                c, p = hwmon_read(dev.index)
                if c is not None:
                    hw.append((c, p))
                time.sleep(0.001)
            loop.drain()
        torch.cuda.synchronize(dev)
        t0, t1, r0, r1 = (int(v) for v in out.tolist())
        return (t1 - t0) / max(r1 - r0, 1) * khz.value * 1e-6, n + 2 * SETTLE_STEPS if busy else 0

    hw = []
    idle, _ = probe(False)
    busy, n = probe(True)
    res = {"shader_ghz_under_this_load": round(busy, 3), "shader_ghz_idle": round(idle, 3), "probe_ms": ms, "launches_beside_the_probe": n,
           "how": "dcx_debug_clock_probe on a side stream: (s_memtime ticks) / (s_memrealtime ticks) x its rate, one wave, beside the loop's launches"}
    if hw:
        pw = [p for _, p in hw if p is not None]
        res["hwmon"] = {"sclk_mhz": round(sum(c for c, _ in hw) / len(hw), 1), "power_w": round(sum(pw) / len(pw), 1) if pw else None,
                        "note": "amdgpu hwmon freq1_input / power1 read while the same launches ran; what freq1_input reports differs between leases"}
    return res


def graph_replay_ms(loop, dev, G=16, reps=40):
    """milliseconds per launch when G launches of the loop's sweep are captured ONCE in a HIP graph and replayed (no host work
    per launch): what the kernel's own phases cost, apart from the host's launch path.  Small launches only
    (a 10 us kernel behind a 5 - 8 us host call is partly host-bound in the eager loop)."""
    main = torch.cuda.Stream(dev)
    with torch.cuda.stream(main):
        for _ in range(4):
            loop._launch(loop.local[0])   # the model's split scratch for this stream exists before the capture
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main, capture_error_mode="thread_local"):
        for _ in range(G):
            loop._launch(loop.local[0])
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(main):
        for _ in range(reps):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / (reps * G)
    del g
    return ms


def two_stream_us(loop, dev, n=200):
    """microseconds per launch when INDEPENDENT launches of the loop's sweep alternate over two streams (launch i + 1's first blocks
    start while launch i's last ones drain) against the same launches in order on one stream: what the ~10 us a B = 65536 launch
    spends outside its steady-state sweep - first FK chains, last fold + J^T, clock ramp - cost a caller that has
    independent batches, and gets back by overlapping them.  A side measurement: `value` is the one-stream loop (a kernel's own
    duration is not defined while two of them share the chip; tools/two_stream_probe.py is the long form)."""
    import ctypes as Ct
    w = loop.w
    B, C, dof = w["B"], w["C"], w["dof"]
    outs = [(torch.empty((B, C), device=dev), torch.empty((B, dof), device=dev)) for _ in range(2)]
    res = {}
    for ns in (1, 2):
        streams = [torch.cuda.Stream(dev) for _ in range(ns)]

        def run(k):
            for i in range(k):
                o, g = outs[i % 2]
                loop._lib.check(loop.lib.dcx_score_grad(w["model"]._h, loop.qp, B, None, Ct.c_void_p(o.data_ptr()), Ct.c_void_p(g.data_ptr()),
                                                        Ct.c_void_p(streams[i % ns].cuda_stream)))
        run(n)
        torch.cuda.synchronize(dev)
        run(int(SETTLE_MS * 1e-3 / 90e-6))
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run(n)
        torch.cuda.synchronize(dev)
        res[ns] = (time.perf_counter() - t0) / n * 1e6
    return {"unit": "us per launch (wall / launches)", "launches": n, "one_stream": round(res[1], 2), "two_streams": round(res[2], 2),
            "gain": round(res[1] / res[2], 3),
            "what": "independent headline launches alternating over two HIP streams against the same launches on one stream"}


def score_only_us(w, dev, n=400):
    """The same batch through dcx_score alone (DiffCo.score / is_collision: no gradient): microseconds per launch, events on the
    launches' stream after the clock has settled.  A side measurement: the headline metric counts score + gradient."""
    m, q = w["model"], w["q"]
    for _ in range(1500):
        m.score_raw(q)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        m.score_raw(q)
    e1.record()
    torch.cuda.synchronize(dev)
    us = e0.elapsed_time(e1) / n * 1e3
    return {"us_per_launch": round(us, 2), "M_scores_per_s": round(w["B"] / us, 1), "launches": n}


def hess_us(w, dev, n=100):
    """dcx_score_hess on the same model (trust-constr's constraint Hessian, analytic second derivatives): microseconds
    per call for a few batch sizes, events on the calls' stream; `lanes_form_us` = the same batch through the other form of the kernel
    (one lane per (configuration, direction) sweeps the supports; the shipped rule takes the moments form from B = 1024)."""
    from diffco_amd import _lib
    lib = _lib.load()
    m = w["model"]
    out = {"unit": "us per call", "what": "gradient + Hessian [B, dof, dof] of the score, q and outputs on the GPU"}

    def timed(q, up):
        for _ in range(30):
            m.score_hess_raw(q, up)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            m.score_hess_raw(q, up)
        e1.record()
        torch.cuda.synchronize(dev)
        return round(e0.elapsed_time(e1) / n * 1e3, 2)

    for B in (256, 1024, 8192, 65536):
        if B > w["B"]:
            continue
        q = w["q"][:B].contiguous()
        up = torch.ones((B, w["C"]), device=dev)
        r = {"us": timed(q, up)}
        if B >= 1024:
            try:
                _lib.check(lib.dcx_debug_set(b"hess_form", 0))
                r["lanes_form_us"] = timed(q, up)
            finally:
                lib.dcx_debug_set(b"hess_form", -1)
        out[f"B{B}"] = r
    return out


def cold_launch_us(loop, dev, n=20, idle_s=1.0, rounds=3):
    """microseconds per launch of the FIRST `n` launches after `idle_s` seconds of an idle GPU (HIP events on the launch stream),
    `rounds` times: the clocks have dropped, the first launches run below the settled rate (tools/clock_ramp.py,
    profiles/r03_clock_ramp.txt: 102 us against 85).  Beside it the next `n` launches and the same after a settle phase."""
    res = []
    for _ in range(rounds):
        loop.drain()
        torch.cuda.synchronize(dev)
        time.sleep(idle_s)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        loop.run(n)
        ev[1].record()
        loop.run(n)
        ev[2].record()
        loop.drain()
        torch.cuda.synchronize(dev)
        res.append((ev[0].elapsed_time(ev[1]) / n * 1e3, ev[1].elapsed_time(ev[2]) / n * 1e3))
    loop.run(int(SETTLE_MS / max(res[-1][1] * 1e-3, 1e-3)))
    loop.drain()
    torch.cuda.synchronize(dev)
    # (three samples of n launches, the middle one: a single sample of 20 launches once read 649 us per launch under the test suite -
    # something else had the GPU, or the host stalled between two launches)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    for k in range(3):
        loop.run(n)
        ev[k + 1].record()
    loop.drain()
    torch.cuda.synchronize(dev)
    settled = sorted(ev[k].elapsed_time(ev[k + 1]) / n * 1e3 for k in range(3))[1]
    return {"unit": "us per launch", "launches": n, "idle_s": idle_s,
            "first": [round(a, 2) for a, _ in res], "next": [round(b, 2) for _, b in res],
            "mean_first": round(sum(a for a, _ in res) / len(res), 2), "settled": round(settled, 2),
            "what": "the headline launch right after the GPU sat idle: mean of the first / next 20 launches, and 20 launches behind "
                    f"{SETTLE_MS:.0f} ms of load (what `value` is measured at)"}


def measure(loop, steps, warmup, dev, multi):
    """W untimed warm-up steps, then exactly `steps` steps bracketed by barrier + synchronize; max over ranks.
    Returns (wall seconds, average launch-to-launch kernel milliseconds from HIP events on the launch stream, number of
    untimed settle steps issued before the warm-up)"""
    # Settle first.  (1) A freshly built model / process group pays one-off costs in its first launches (lazy RCCL buffers
    # for a new message size ...).  (2) The GPU's clocks ramp for ~50 ms of sustained load after an idle period: the headline
    # launch takes 102 us in its first 100 launches, 93, 88, 86 in the next hundreds and 85.1-85.3 us from 45 ms on
    # (tools/clock_ramp.py, profiles/r03_clock_ramp.txt); a 20-step run behind 5 warm-up steps measures the ramp, not the
    # kernel.  So the loop first keeps the GPU busy for SETTLE_MS with untimed launches - the state a planner's loop is
    # in - and only then runs the W warm-up steps and the K timed ones.  The count is agreed across ranks (they issue
    # the same collectives).
    loop.run(SETTLE_STEPS)
    loop.drain()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    loop.run(SETTLE_STEPS)
    loop.drain()
    e1.record()
    torch.cuda.synchronize(dev)
    est_ms = max(e0.elapsed_time(e1) / SETTLE_STEPS, 1e-3)
    n_settle = int(min(max(SETTLE_MS / est_ms, 0), 20000))
    if multi:
        t = torch.tensor([n_settle], device="cpu" if SAME_GPU else dev, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n_settle = int(t.item())
    loop.run(n_settle)
    loop.drain()
    torch.cuda.synchronize(dev)
    loop.run(warmup)
    loop.drain()
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    loop.run(steps, timed=True)
    e1.record()      # closes the kernels on the launch stream (HIP events, same stream as the launches)
    loop.drain()
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    kern_ms = e0.elapsed_time(e1) / steps
    if multi:
        tt = torch.tensor([wall, kern_ms], device="cpu" if SAME_GPU else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, kern_ms = float(tt[0]), float(tt[1])
    return wall, kern_ms, 2 * SETTLE_STEPS + n_settle


def primary_line(args, w, loop, world, multi, ranks_reported, wall, kern_ms, n_settle, gather_ms, ge, is_traj, cpu_base, cpu_torch):
    """the JSON line of the timed region (everything but the side measurements `variants` / `configs`)"""
    name = args.workload
    B, C, dof = w["B"], w["C"], w["dof"]
    value = ge * args.steps / wall / 1e6
    F = flops_per_eval(w["D"], C, w["S"])
    ach_tf = F * B / (kern_ms * 1e-3) / 1e12
    ach_gbs = bytes_per_eval(dof, C) * B / (kern_ms * 1e-3) / 1e9
    pmc = load_profile_json(f"pmc_{name}.json")
    if pmc is not None and pmc.get("batch_per_gpu") not in (None, B):
        pmc = None   # (the counters were collected at another batch size: bytes per launch do not carry over)
    mfc = load_profile_json("mfma_contractions.json") or {}
    forms = mfc.get("forms") or {}
    head_form = next((v for k, v in forms.items() if k.startswith("headline")), {})
    this_form = next((v for k, v in forms.items() if k.split(" ")[0] == name), head_form)
    mfma_on = os.environ.get("DCX_MFMA", "") not in ("", "0", "-1") and w["D"] <= 16 and w["D"] % 2 == 0 and C in (1, 5, 8) and not is_traj
    gather_txt = {"graph": "RCCL all-gather of the scores of EVERY call beside the next call's sweep, sweep + gather captured "
                           "in a HIP graph (8 calls per replay)",
                  "per-call": "RCCL all-gather of the scores after EVERY call, in order on the launch stream",
                  "overlapped": "RCCL all-gather of the scores after every call, beside the next call's sweep",
                  "bucketed": f"RCCL all-gather of the scores every {GATHER_EVERY} calls, overlapped",
                  "none": "no gather (sharded consumer)",
                  "summaries": "restarts sharded; only per-restart summaries + candidate paths gathered, once"}[loop.gather]
    out = {
        "metric": "million collision-score+grad evals/sec, 7-DoF FK-kernel, 2k supports",
        "value": round(value, 3), "unit": "M evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 5), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        # untimed launches that keep the GPU busy before the W warm-up steps (clock ramp: tools/clock_ramp.py)
        "settle_ms": SETTLE_MS, "settle_steps": n_settle,
        "config": {"workload": f"{w['name']}: {w['text']}", "batch_per_gpu": B, "global_batch": ge,
                   "supports": w["S"], "features": w["D"], "classes": C,
                   "parallelism": f"batch-sharded x{world}, model replicated" + ("" if not multi else ", " + gather_txt),
                   "launches_per_step": round(-(-args.steps // 192) / max(args.steps, 1), 4) if is_traj else 1},
        "roofline": {"bound": "valu", "achieved": round(ach_tf, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ach_tf / PEAK_FP32_TFLOPS, 4),
                     "traffic": None if pmc is None else pmc.get("hbm_bytes_per_launch"),
                     "traffic_source": None if pmc is None else f"profiles/pmc_{name}.json (rocprofv3 --pmc passes of an earlier run of "
                                                                 "this command, calibrated; a constant, not an observation of this run)",
                     "kernel": "dcx::traj_fused_kernel<D,KF,MAXT,XF>" if is_traj else "dcx::score_kernel<D,KF,C,MODE,MAXT,MF,XF>",
                     "sweep_form": ("expanded with x.s^T on the matrix cores (XM: bf16x3 split operands on v_mfma_f32_16x16x32_bf16; DCX_XM=1)"
                                    if (os.environ.get("DCX_XM", "") not in ("", "0", "-1") and w["kspec"][0] == 1 and w["kspec"][1] == 1.0
                                        and C == 1 and w["D"] <= 16 and w["D"] % 2 == 0 and not is_traj and not mfma_on
                                        and os.environ.get("DCX_XF", "") != "0") else
                                    "expanded (XF: d2 = |x|^2+|s|^2-2x.s, gX = x*sum(c)-sum(c s); 20 VALU/pair at D=12)"
                                    if (w["kspec"][0] == 1 and w["kspec"][1] == 1.0 and w["D"] + C + (C > 1) + 1 <= 38
                                        and os.environ.get("DCX_XF", "") != "0" and not mfma_on)
                                    else "expanded where libdcx's rule admits it (RQ2 behind an FK transform, gamma*max|s-c|^2 <= 32: it does for "
                                         "the Baxter workloads; 22 VALU/pair at D=12, C=5), else direct with the RQ constants folded"
                                    if (w["kspec"][0] == 0 and w["kspec"][2] == 2.0 and w["rob_name"] is not None
                                        and os.environ.get("DCX_XF", "") != "0" and not mfma_on)
                                    else "direct (differences; RQ2: constants folded, 15 VALU/pair at D=6)"),
                     "kernel_ms": round(kern_ms, 5), "flops_per_eval": F,
                     "note": "fp32 VALU bound (peak == fp32 MFMA peak 157.3 TFLOP/s); algorithmic flops "
                             "S*(5D+4C+6)+800 per eval (SURVEY.md §8d)",
                     "hbm": {"achieved": round(ach_gbs, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": round(ach_gbs / PEAK_HBM_GBS, 5),
                             "bytes_per_eval": bytes_per_eval(dof, C)},
                     # the matrix cores: what the default path issues (nothing) and what the measured MFMA forms of
                     # this path cost - the north star's K[B,S].W[S,C] contraction at C >= 4 included
                     # (profiles/mfma_contractions.json <- profiles/r03_mfma_ab.txt)
                     "mfma": {"used": bool(mfma_on),
                              "instructions_per_launch": this_form.get("instructions_per_launch") if mfma_on else 0,
                              "busy_frac": this_form.get("busy_frac") if mfma_on else 0.0,
                              "contraction": ("K[B,S].W[S,C] (C >= 4) and the (configurations x supports).(supports x features) "
                                              "gradient fold on v_mfma_f32_16x16x4_f32; upstream.W^T measured in isolation"),
                              "measured_variant": {**{k: head_form.get(k) for k in (
                                  "instructions_per_launch", "busy_frac", "mfma_flops_per_launch", "kernel_us_mfma_form",
                                  "kernel_us_valu_form")}, "verdict": mfc.get("verdict"), "source": mfc.get("source")},
                              "forms": mfc.get("forms"),
                              "contractions_in_isolation": mfc.get("contractions_in_isolation_8_waves_per_simd"),
                              "coissue": mfc.get("coissue")}},
    }
    if multi:
        out["multi"] = {"ranks": ranks_reported, "backend": "gloo, all ranks on cuda:0 (rehearsal)" if SAME_GPU else "nccl (RCCL)", "gather": loop.gather,
                        "gather_ms": None if gather_ms is None else round(gather_ms, 5),
                        "gather_bytes_per_call": None if is_traj else
                        world * B * C * 4 * (GATHER_EVERY if loop.gather == "bucketed" else 1)}
    out["cpu_baseline"] = cpu_base
    if cpu_torch:
        out["cpu_baseline_torch"] = cpu_torch
    return out


def solve_times(dev):
    """K x = y for the Polyharmonic(1) kernel matrix of S supports in 24 features, as fit_poly builds it (dcx_kernel_matrix):
    dcx_solve (one launch, diffco_amd._ops.solve) and torch.linalg.solve (hipSOLVER) on the SAME device matrix, milliseconds
    of wall time per call in a loop of 10, device-synchronised; the residual max|K x - y| in float64 beside each"""
    from diffco_amd import _ops
    out = {"unit": "ms", "what": "S x S solve of fit_poly, one right-hand side, fp32 in / out"}
    for S in (438, 2000):
        g = torch.Generator().manual_seed(S)
        feats = torch.rand((S, 24), generator=g).to(dev)
        y = torch.sign(torch.randn(S, generator=g)).to(dev)
        K = _ops.kernel_matrix(1, 1.0, 1.0, feats, feats)   # DCX_K_POLY, k = 1, epsilon = 1
        res = {}
        for key, fn in (("dcx_solve", lambda: _ops.solve(K, y)), ("hipsolver", lambda: torch.linalg.solve(K, y))):
            x = fn()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(10):
                x = fn()
            torch.cuda.synchronize(dev)
            res[key] = round((time.perf_counter() - t0) / 10 * 1e3, 4)
            res[key + "_residual"] = float((K.double() @ x.double() - y.double()).abs().max())
        out[f"S{S}"] = res
    return out


def api_latency(dev, n=200):
    """What a caller of the drop-in Python API waits for per call, microseconds of wall time in a loop of `n` (device
    synchronised at the end, so consecutive calls overlap as they do in an optimiser's loop): `DiffCo.poly_score(q)` without
    and with `torch.autograd.grad` behind it (the reference optimisers' call pattern, optim.py:88-101), the non-autograd
    `ScoreModel.score_and_grad`, and the raw C-ABI call on ready device buffers - Baxter, 2000 supports, q resident on the
    GPU.  `torch_autograd_floor` = forward + autograd.grad of `(q * 2).sum()` alone: what torch's engine costs on this host
    without any of this package's code in the graph (tools/api_latency.py is the long form)."""
    from diffco_amd import kernel, model
    from diffco_amd.kernel_perceptrons import DiffCo
    rob = model.BaxterLeftArmFK()
    lim = rob.limits
    g = torch.Generator().manual_seed(0)
    S = 2000
    sq = (torch.rand((S, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).to(dev)
    dc = DiffCo(transform=rob.fkine)
    dc.support_points, dc.support_transformed = sq, rob.fkine(sq)
    dc.rbf_kernel, dc.rbf_nodes = kernel.Polyharmonic(1, 1.0), torch.randn(S, generator=g).to(dev)

    def timeit(fn):
        best = 1e30
        for _ in range(2):   # (the better of two loops: the host of a GPU box is not quiet)
            for _ in range(20):
                fn()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize(dev)
            best = min(best, (time.perf_counter() - t0) / n * 1e6)
        return round(best, 2)

    qe = torch.rand((50, 7), device=dev, requires_grad=True)
    out = {"unit": "us per call", "what": "DiffCo.poly_score on a Baxter checker with 2000 supports, q on the GPU",
           "torch_autograd_floor": timeit(lambda: torch.autograd.grad((qe * 2.0).sum(), qe))}
    for B in (20, 50, 256, 4096):
        q = (torch.rand((B, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).to(dev)
        qg = q.clone().requires_grad_(True)
        m = dc._poly_fused.model(dc.transform, dc.rbf_kernel, dc.support_transformed, dc.rbf_nodes, dev)

        def fwd():
            with torch.no_grad():
                return dc.poly_score(q)

        def fwd_bwd():
            return torch.autograd.grad(dc.poly_score(qg).sum(), qg)

        # (the calls that do not touch torch's autograd engine first: its device thread stays warm for a while after a backward)
        r = {"raw": timeit(lambda: m.score_grad_raw(q)), "score_and_grad": timeit(lambda: m.score_and_grad(q)), "fwd": timeit(fwd)}
        r["fwd_bwd"] = timeit(fwd_bwd)
        out[f"B{B}"] = r
    return out


def escape_latency(dev, n=100):
    """The escape loop next to the path (SURVEY 8f-2's variant, reference scripts/escape.py:19-38) with the options of
    scripts/compare_sampling.py:177-195 - one configuration, at most three Adam steps at lr 0.2, wrap2pi, last configuration only -
    on a Baxter checker with 2000 supports: microseconds of wall time per escape for the fused form (`dcx_escape_adam`: one
    library call, one read-back) and for the reference's Python loop on the same HIP score (autograd + torch.optim.Adam +
    a host decision per step), on a start that takes all three steps; and 65536 independent loops of 20 steps
    (`optim_escape_batch`, stopped loops compacted out of the sweep every 4 steps).  tools/escape_bench.py is the long form."""
    from diffco_amd import kernel, model, utils
    from diffco_amd.escape import OptimSampler
    from diffco_amd.kernel_perceptrons import DiffCo
    rob = model.BaxterLeftArmFK()
    lim = rob.limits
    g = torch.Generator().manual_seed(0)
    S = 2000
    sq = (torch.rand((S, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).to(dev)
    dc = DiffCo(transform=rob.fkine)
    dc.support_points, dc.support_transformed = sq, rob.fkine(sq)
    dc.rbf_kernel, dc.rbf_nodes = kernel.Polyharmonic(1, 1.0), (torch.randn(S, generator=g) * 0.02 + 0.001).to(dev)
    starts = (torch.rand((65536, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).to(dev)
    s0 = dc.poly_score(starts[:4096]).reshape(-1)
    margin = float(s0.median()) - 0.2
    one = starts[int(torch.argmax(s0))][None].clone()
    # (a margin no step reaches: the loop always takes its three steps, whatever the lease's rounding)
    opts = {"N_WAYPOINTS": 3, "safety_margin": -1e3, "lr": 0.2, "record_freq": None, "post_transform": utils.wrap2pi}
    fused = OptimSampler(rob, dc.poly_score, opts)
    host = OptimSampler(rob, dc.poly_score, dict(opts, post_transform=lambda x: utils.wrap2pi(x)))

    def timeit(fn, reps):
        fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / reps
    out = {"unit": "us per escape", "what": "OptimSampler.optim_escape, Baxter checker with 2000 supports, one configuration, 3 Adam steps, wrap2pi",
           "evaluations": fused.optim_escape(one)[1], "routes": None}
    out["fused"] = round(timeit(lambda: fused.optim_escape(one), n) * 1e6, 1)
    out["host_loop"] = round(timeit(lambda: host.optim_escape(one), max(10, n // 5)) * 1e6, 1)
    out["routes"] = [fused.last_route, host.last_route]
    batch = OptimSampler(rob, dc.poly_score, dict(opts, N_WAYPOINTS=20, lr=5e-2, safety_margin=margin))
    tb = timeit(lambda: batch.optim_escape_batch(starts), 5)
    out["batch_65536x20"] = {"ms": round(tb * 1e3, 3), "M_escapes_per_s": round(65536 / tb / 1e6, 2)}
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: this process BECOMES `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> bench.py <the same arguments>` (exec: same
    pid, same stdout, the launcher's exit code is the job's).  Rank 0 of the job it starts owns the line (LineKeeper).  An
    external launcher is still accepted: WORLD_SIZE in the environment means the ranks already exist."""
    import socket
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.pop("MASTER_PORT", None)   # (the launcher exports its own to the ranks)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def main():
    if len(sys.argv) == 6 and sys.argv[1] == "--torch-cpu-leg":
        return _torch_cpu_leg(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the workload's per-GPU batch on every rank; strong: its fixed global batch divided by the ranks")
    ap.add_argument("--gather", default="per-call", choices=["graph", "per-call", "overlapped", "bucketed", "none"],
                    help="N>1: per-call (default) = every call's scores all-gathered in order on the launch stream (always works, and "
                         "the fastest form measured on one rank, profiles/r03_bench_forcedist.jsonl); graph = beside the NEXT call's "
                         "sweep, eight calls captured in one HIP graph (falls back to per-call if the collective cannot be captured); "
                         "overlapped = eager, beside the next sweep; bucketed = every 4 calls; none = sharded consumer.  The other "
                         "forms are measured briefly as `variants` AFTER the primary line is safe, graph last")
    ap.add_argument("--no-gather", action="store_true", help="same as --gather none")
    ap.add_argument("--no-variants", action="store_true", help="N>1: skip the short runs of the other variants")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="headline, N=1: skip the short measurements of the other BASELINE configurations (`configs` in the line)")
    ap.add_argument("--force-dist", action="store_true",
                    help="exercise the N>1 code path (process group + all-gather) even with one rank")
    args = ap.parse_args()
    if args.no_gather:
        args.gather = "none"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)   # does not return: this process becomes the launcher of the N ranks

    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL writes its version banner to stdout
    # when the first communicator comes up), so file descriptor 1 is pointed at stderr for the whole run and the JSON
    # line goes to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from diffco_amd import _lib
    _lib.require_gpu()
    if SAME_GPU:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    keeper = LineKeeper(real_stdout) if rank == 0 else None

    # The CPU baseline (rank 0's host cores) runs FIRST, before the process group exists: the other ranks wait in the TCP
    # rendezvous, not inside a collective kernel, and the primary line below is complete the moment the timed region ends.
    cpu_base, cpu_torch = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # (N = 1 only: a launcher gives its ranks OMP_NUM_THREADS = 1)
        w0 = make_workload(args.workload, args.batch or min(WORKLOADS[args.workload][4], 65536), dev, seed=rank)
        cpu_base = cpu_baseline(w0)
        cpu_torch = torch_cpu_baseline(w0)
        del w0
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if SAME_GPU:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ranks_reported = dist.get_world_size() if multi else 1

    name = args.workload
    is_traj = name.startswith("cfg5")
    weak_B, strong_global = WORKLOADS[name][4], WORKLOADS[name][5]
    unit = TRAJ_W if is_traj else 1  # config #5 shards whole restarts

    def per_gpu_batch(scaling):
        if args.batch:
            return args.batch
        if scaling == "weak":
            return weak_B
        from diffco_amd.sharded import shard_bounds
        lo, hi = shard_bounds(strong_global // unit, rank, world)
        return (hi - lo) * unit

    def build(scaling, gather):
        B = per_gpu_batch(scaling)
        w = make_workload(name, B, dev, seed=rank)
        if is_traj:
            n_total = (B // TRAJ_W) * world if (scaling == "weak" or args.batch) else strong_global // TRAJ_W
            return w, TrajLoop(w, dev, world, n_total, None)
        return w, ScoreLoop(w, dev, world, gather if multi else "none")

    def global_evals(scaling, w):
        if args.batch or scaling == "weak":
            return world * w["B"]
        return strong_global

    w, loop = build(args.scaling, args.gather)
    wall, kern_ms, n_settle = measure(loop, args.steps, args.warmup, dev, multi)
    gather_ms = loop.gather_ms()
    B, C, dof = w["B"], w["C"], w["dof"]

    out = None
    if rank == 0:
        out = primary_line(args, w, loop, world, multi, ranks_reported, wall, kern_ms, n_settle, gather_ms, global_evals(args.scaling, w),
                           is_traj, cpu_base, cpu_torch)
        # the clock the SIMDs really had under this loop's load (measured in the kernel, right AFTER the timed region: nothing
        # samples anything while the K steps are timed), the fraction against the peak at that clock, the driver's sensors beside it
        try:
            # (N = 1 only: the probe's busy loop would issue this rank's collectives without its peers)
            ck = clock_under_load(loop, dev) if not (SAME_GPU or multi) else None
        except Exception as exc:  # noqa: BLE001  (a side measurement)
            ck = {"error": f"{type(exc).__name__}: {exc}"[:160]}
        if ck is not None:
            out["roofline"]["clock"] = ck
            if ck.get("shader_ghz_under_this_load"):
                out["roofline"]["frac_at_measured_clock"] = round(out["roofline"]["frac"] * 2.4 / ck["shader_ghz_under_this_load"], 4)
                out["roofline"]["clock_note"] = ("peak 157.3 TFLOP/s assumes 2.4 GHz; frac_at_measured_clock = frac x 2.4 / shader_ghz_under_this_load. "
                                                 "It can exceed 1: `achieved` counts ALGORITHMIC flops (S*(5D+4C+6)+800 per evaluation) and the "
                                                 "expanded-form sweep executes ~0.8 of them (profiles/r05_clock_under_load.txt)")
        keeper.primary(out)   # from here on the driver gets a line whatever happens below
    _fault("after_primary")
    # ... also if a side measurement HANGS (a collective whose peers took another path, a capture that never returns): every rank
    # arms a timer when its primary numbers are safe; when it fires the rank leaves at once with exit code 0, and rank 0's
    # keeper prints the best line it was given (round 5; an abort was already covered, a hang ran into the driver's own timeout).
    side_budget = float(os.environ.get("DCX_BENCH_SIDE_BUDGET_S", "240"))

    def _leave():
        sys.stderr.write(f"bench: side measurements exceeded {side_budget:.0f} s; leaving with the line measured so far\n")
        sys.stderr.flush()
        if keeper is not None:
            keeper.close()
        os._exit(0)
    import threading
    side_timer = threading.Timer(side_budget, _leave)
    side_timer.daemon = True
    side_timer.start()
    if os.environ.get("DCX_BENCH_FAULT", "") == "hang":
        sys.stderr.write("bench: injected hang in the side measurements\n")
        time.sleep(10 ** 6)

    variants = None
    candidates = []   # (wall, gather form, workload, loop, kernel ms, settle steps) of the forms that may be promoted
    if multi and not args.no_variants:
        # the other ways to run N > 1, measured briefly in the same job (primary numbers above are untouched)
        vs, vw = max(48, args.steps // 4), max(8, args.warmup // 2)
        variants = {}
        others = [("other_scaling", "strong" if args.scaling == "weak" else "weak", args.gather if args.gather != "graph" else "per-call")]
        if not is_traj:
            # the captured form LAST: a collective that cannot be captured can take the process down from c10d's watchdog
            # thread (see ScoreLoop._capture); by then every other variant is measured and the primary line is with the keeper
            others += [(f"gather_{g}", args.scaling, g) for g in ("per-call", "none", "bucketed", "overlapped", "graph") if g != args.gather]
        for key, sc, ga in others:
            # A variant is a side measurement: if one fails (the same way on every rank: an allocation, an argument), it
            # is recorded as failed and the primary line above still goes out.
            try:
                w2, l2 = build(sc, ga)
                if sc == args.scaling and ga in PROMOTABLE and args.gather in PROMOTABLE and not is_traj:
                    # A form that hands the consumer the gathered scores of EVERY call, like the primary one: measured exactly
                    # as the primary line was - W warm-up steps, ONE timed region of exactly K steps, max over ranks - because
                    # the fastest of these that completes becomes the line's `value` (promotion, below)
                    wl, km, ns = measure(l2, args.steps, args.warmup, dev, multi)
                    nsteps = args.steps
                    if l2.gather == ga:   # (a captured form that fell back to per-call is that, not a candidate)
                        candidates.append((wl, ga, w2, l2, km, ns))
                else:
                    # best of two short runs: with a process group alive, a ~100 ms stall of unknown origin (seen in the no-gather
                    # variant too) occasionally lands inside one 48-step run; the primary measurement above is a single run, as agreed
                    wl, km = min(measure(l2, vs, vw, dev, multi)[:2], measure(l2, vs, vw, dev, multi)[:2])
                    nsteps = vs
                ge = global_evals(sc, w2)
                variants[key] = {"scaling": sc, "gather": l2.gather, "value": round(ge * nsteps / wl / 1e6, 3),
                                 "ms_per_step": round(wl / nsteps * 1e3, 5), "kernel_ms": round(km, 5), "steps": nsteps,
                                 "global_batch": ge, "batch_per_gpu": w2["B"],
                                 "gather_ms": None if l2.gather_ms() is None else round(l2.gather_ms(), 5)}
                if not (candidates and candidates[-1][3] is l2):
                    del w2, l2
            except Exception as exc:  # noqa: BLE001
                variants[key] = {"scaling": sc, "gather": ga, "error": f"{type(exc).__name__}: {exc}"[:200]}

    # The other BASELINE configurations, measured briefly in the same job on the same box (N = 1, headline run only): the
    # driver's bench line then carries a number for every entry of BASELINE.json.configs that runs on a GPU.
    configs = None
    if world == 1 and not multi and name == "headline" and not args.batch and not args.no_configs:
        configs = {}
        for cname, csteps in (("cfg2", 200), ("cfg2_panda", 200), ("cfg3", 200), ("cfg3_b65536", 200), ("cfg3_poly", 200), ("cfg4", 12),
                              ("cfg5", 200), ("cfg5_shard32", 200), ("cfg5_c5", 200), ("headline_rq", 100)):
            try:
                wname = cname.split("_shard")[0].split("_b")[0]
                cbatch = {"cfg5_shard32": 32 * TRAJ_W, "cfg3_b65536": 65536}.get(cname, WORKLOADS[wname][4])
                cw = make_workload(wname, cbatch, dev, seed=rank)
                if wname.startswith("cfg5"):
                    cl = TrajLoop(cw, dev, 1, cbatch // TRAJ_W, None)
                else:
                    cl = ScoreLoop(cw, dev, 1, "none")
                cwall, ckm = measure(cl, csteps, 10, dev, False)[:2]
                cF = flops_per_eval(cw["D"], cw["C"], cw["S"])
                ctf = cF * cw["B"] / (ckm * 1e-3) / 1e12
                configs[cname] = {"workload": cw["text"], "batch": cw["B"], "steps": csteps,
                                  "value": round(cw["B"] * csteps / cwall / 1e6, 3), "unit": "M evals/s",
                                  "ms_per_step": round(cwall / csteps * 1e3, 5), "kernel_ms": round(ckm, 5),
                                  "flops_per_eval": cF, "frac": round(ctf / PEAK_FP32_TFLOPS, 4)}
                if cname in ("cfg2", "cfg2_panda", "cfg3", "cfg3_poly"):
                    try:   # the same launch replayed from a HIP graph: the kernel's phases without the host's launch path
                        configs[cname]["graph_ms_per_step"] = round(graph_replay_ms(cl, dev), 5)
                    except Exception as exc:  # noqa: BLE001
                        configs[cname]["graph_ms_per_step"] = f"{type(exc).__name__}: {exc}"[:120]
                del cw, cl
            except Exception as exc:  # noqa: BLE001  (a side measurement never takes the primary line down)
                configs[cname] = {"error": f"{type(exc).__name__}: {exc}"[:200]}

    # What ONE GPU's numbers already say about 8: under strong scaling a rank runs 1/8 of the batch, so the speed-up
    # is at most t(1 GPU, B) / t(1 GPU, B / 8) - before any gather.  Small launches are latency-bound, hence far from 8.
    strong_bound = None
    if configs is not None:
        def _ms(k):
            return (configs.get(k) or {}).get("ms_per_step")
        strong_bound = {"what": "t(1 GPU, B) / t(1 GPU, B / 8) measured in this job: the most 8 GPUs can gain at FIXED total work, before the gather",
                        "cfg3_65536_over_8": None if not (_ms("cfg3_b65536") and _ms("cfg3")) else round(_ms("cfg3_b65536") / _ms("cfg3"), 2),
                        "cfg5_256_restarts_over_8": None if not (_ms("cfg5") and _ms("cfg5_shard32")) else round(_ms("cfg5") / _ms("cfg5_shard32"), 2),
                        "weak": "at fixed work per GPU (bench.py's default, --scaling weak) every rank runs the single-GPU launch: bound 8"}

    # the caller next to the path: fit_poly's S x S solve, one launch (dcx_solve), the library route beside it
    callers = None
    if configs is not None:
        callers = {}
        try:   # what a planner's FIRST calls after a pause see: the settled `value` is the other end
            callers["headline_cold_us"] = cold_launch_us(loop, dev)
        except Exception as exc:  # noqa: BLE001
            callers["headline_cold_us"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        try:   # ... and what overlapping independent launches gives back (round 6)
            callers["headline_two_streams_us"] = two_stream_us(loop, dev)
        except Exception as exc:  # noqa: BLE001
            callers["headline_two_streams_us"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        if not is_traj:
            try:   # the batch's scores alone (dcx_score: the checker's score() / is_collision)
                callers["headline_score_only"] = score_only_us(w, dev)
            except Exception as exc:  # noqa: BLE001
                callers["headline_score_only"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        if not is_traj:
            try:   # second derivatives of the same model (row f4)
                callers["score_hess_us"] = hess_us(w, dev)
            except Exception as exc:  # noqa: BLE001
                callers["score_hess_us"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        try:
            callers["fit_poly_solve"] = solve_times(dev)
        except Exception as exc:  # noqa: BLE001
            callers["fit_poly_solve"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        try:   # the Python boundary itself
            callers["poly_score_us"] = api_latency(dev)
        except Exception as exc:  # noqa: BLE001
            callers["poly_score_us"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        try:   # the escape loop (round 5 widening)
            callers["escape_us"] = escape_latency(dev)
        except Exception as exc:  # noqa: BLE001
            callers["escape_us"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}

    # Promotion, first half (every rank: the walls are maxima over the ranks, so all of them pick the same winner): the fastest
    # candidate is MEASURED AGAIN, once - the minimum of three noisy single runs is biased low, a fresh run of the
    # winner is not - and it is promoted only if that run still beats the primary one.
    best = min(candidates, key=lambda c: c[0]) if candidates else None
    if best is not None and best[0] < wall:
        try:
            rw, rkm, rns = measure(best[3], args.steps, args.warmup, dev, multi)
            best = (rw, best[1], best[2], best[3], rkm, rns)
        except Exception:  # noqa: BLE001  (a side measurement)
            best = None
    if rank == 0:
        if multi:
            # Promotion: the line's `value` is the FASTEST form that completed among those that deliver every
            # call's gathered scores (per-call in order, eager overlapped, captured graph) - each timed once over exactly K steps
            # behind W warm-up steps, max over ranks; the winner's number is the one of its SECOND run (see above).  The per-call
            # form was parked first (`keeper.primary`), so a form that takes the process down costs nothing; `multi.primary` keeps
            # its numbers, `multi.gather` names the form `value` is of.
            first = {"gather": loop.gather, "value": out["value"], "ms_per_step": out["ms_per_step"],
                     "kernel_ms": out["roofline"]["kernel_ms"], "gather_ms": out["multi"]["gather_ms"]}
            if best is not None and best[0] < wall:
                bw, bga, w2, l2, bkm, bns = best
                clk_first = out["roofline"].get("clock")
                out = primary_line(args, w2, l2, world, multi, ranks_reported, bw, bkm, bns, l2.gather_ms(), global_evals(args.scaling, w2),
                                   is_traj, cpu_base, cpu_torch)
                variants[f"gather_{first['gather']}"] = {"scaling": args.scaling, **first, "steps": args.steps,
                                                          "global_batch": global_evals(args.scaling, w), "batch_per_gpu": w["B"]}
                variants.pop(f"gather_{bga}", None)
                wall = bw
                if clk_first is not None:
                    out["roofline"]["clock"] = dict(clk_first, note="measured beside the per-call run (multi.primary)")
                    if clk_first.get("shader_ghz_under_this_load"):
                        out["roofline"]["frac_at_measured_clock"] = round(out["roofline"]["frac"] * 2.4 / clk_first["shader_ghz_under_this_load"], 4)
                        out["roofline"]["clock_note"] = ("frac of the promoted run x 2.4 / the shader clock measured beside the per-call run "
                                                         "(the probe does not run beside collectives)")
            out["multi"]["primary"] = first
            out["multi"]["promoted"] = out["multi"]["gather"] != first["gather"]
            none_ms = ((variants or {}).get("gather_none") or {}).get("ms_per_step")
            # what the gather adds to a step: this line's step time minus the same job's no-gather variant
            out["multi"]["gather_exposed_ms"] = None if none_ms is None else round(wall / args.steps * 1e3 - none_ms, 5)
            if variants is not None:
                out["variants"] = variants
        if configs is not None:
            out["configs"] = configs
            out["strong_bound"] = strong_bound
            out["callers"] = callers
        keeper.final(out)
    side_timer.cancel()
    if multi:
        dist.destroy_process_group()
    if keeper is not None:
        keeper.close()


if __name__ == "__main__":
    main()
