#!/usr/bin/env python3
"""bench.py — throughput of the fused score+grad hot path on MI355X.

Metric (BASELINE.json): million collision-score+grad evaluations per second, 7-DoF FK kernel, 2k supports.
One evaluation = one configuration q[7] -> score[C] and d(sum_c upstream*score)/dq[7] against all S supports.
A "step" = one pass of the hot path (ONE `dcx_score_grad` launch) over one batch of synthetic
configurations already resident in HBM.  With N > 1 ranks every rank runs its own batch (weak scaling)
and the scores are all-gathered over RCCL/xGMI each step, overlapped with the next step's sweep.

    python bench.py                      # 1 GPU, headline workload
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` and `cpu_baseline`.
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
    # name: (robot, kernel (kind,p0,p1), S, C, per-GPU batch, description)
    "headline": ("baxter", (1, 1.0, 1.0), 2000, 1, 65536,
                 "7-DoF Baxter DH chain (D=12), Polyharmonic(1,1), S=2000, C=1 (SURVEY.md §8d headline)"),
    "cfg2": ("baxter", (1, 1.0, 1.0), 1000, 1, 4096, "BASELINE config #2: 7-DoF, FK kernel, 1k supports, batch 4096"),
    "cfg2_panda": ("panda", (1, 1.0, 1.0), 1000, 1, 4096, "config #2 with PandaFK (D=21)"),
    "cfg3": ("baxter", (0, 10.0, 2.0), 2000, 5, 8192, "BASELINE config #3: MultiDiffCo C=5, RQ(10), S=2000, 8192 per GPU"),
    "cfg4": (None, (0, 10.0, 2.0), 10000, 1, 1 << 20, "BASELINE config #4: SE(3) no-FK (D=6), RQ(10), S=10k, 1M configs"),
    # the reference's recommended facade (ForwardKinematicsDiffCo on panda.urdf, tutorial cell 13): URDF tree, 8 dof
    "urdf_panda": ("urdf_panda", (1, 1.0, 1.0), 2000, 1, 65536,
                   "URDF Panda with gripper (DCX_FK_TREE: 8 dof, 9 link origins, D=27), Polyharmonic(1,1), S=2000, C=1"),
    # config #5: a "step" is ONE fused Adam iteration over 256 restarts x 50 waypoints (= 12800 score+grad evals)
    "cfg5": ("baxter", (1, 1.0, 1.0), 2000, 1, 256 * 50,
             "BASELINE config #5: fused Adam trajopt, 7-DoF, 50 waypoints x 256 restarts per GPU, S=2000 "
             "(step = 1 iteration: score+hinge-grad sweep + fused Adam step)"),
}


GATHER_EVERY = 4  # N > 1: steps per all-gather bucket


def flops_per_eval(D, C, S):
    """F_pair = 5D + 4C + 6, F_eval = S*F_pair + 800 (FK + J^T)"""
    return S * (5 * D + 4 * C + 6) + 800


def bytes_per_eval(dof, C):
    """SURVEY.md §8d algorithmic HBM bytes: q in, score out, grad out"""
    return 4 * dof + 4 * C + 4 * dof


def make_workload(name, batch, dev, seed=0):
    from diffco_amd import _fkdesc, _ops, model
    rob_name, kspec, S, C, B, desc_txt = WORKLOADS[name]
    B = batch or B
    g = torch.Generator().manual_seed(0)               # the model (supports, weights) is the same on every rank
    gq = torch.Generator().manual_seed(1000 + seed)    # the batch shard is the rank's own
    if rob_name is None:
        lo = torch.tensor([-10.0] * 3 + [-np.pi] * 3)
        hi = -lo
        desc = _fkdesc.none_desc(6)
    else:
        if rob_name.startswith("urdf_"):
            # joint table stored with the golden FK fixture (tests/golden/fk_urdf_*.npz; no URDF file needed)
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
            import helpers
            rob = helpers.urdf_robot(rob_name)
        else:
            rob = {"baxter": model.BaxterLeftArmFK, "panda": model.PandaFK}[rob_name]()
        lo, hi = rob.limits[:, 0], rob.limits[:, 1]
        desc = rob.fk_desc()
    sup_q = torch.rand((S, len(lo)), generator=g) * (hi - lo) + lo
    W = torch.randn((S, C), generator=g)
    if name == "cfg3":  # 40 % of the entries zeroed per class (mimics deprecated/MultiDiffCo.py:152-153)
        W = W * (torch.rand((S, C), generator=g) >= 0.4)
    q = torch.rand((B, len(lo)), generator=gq) * (hi - lo) + lo
    sup = _ops.fkine(desc, sup_q.to(dev)).reshape(S, -1)
    m = _ops.ScoreModel(desc, *kspec, sup, W.to(dev), device=dev)
    return dict(name=name, text=desc_txt, model=m, desc=desc, kspec=kspec, S=S, C=C, B=B, D=desc.feature_dim,
                dof=desc.dof, q=q.to(dev).contiguous(), sup=sup, W=W, q_cpu=q, rob_name=rob_name, lo=lo, hi=hi)


def traj_state(w, R, Wp, dev):
    """config #5 trajectory state: R restarts of Wp waypoints in joint limits; grad_tol = 0 so no path ever freezes"""
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


def torch_cpu_baseline(w, budget_s=6.0):
    """A torch-CPU restatement of the reference EXPRESSION (cdist -> kernel -> matmul, autograd backward)
    on precomputed features — what the reference does per call minus its FK.  Secondary information."""
    if w["kspec"][0] not in (0, 1):
        return None
    nthr = min(os.cpu_count() or 1, 32)  # more threads than that only slows torch down at this size
    torch.set_num_threads(nthr)
    sup = w["sup"].cpu()
    Wt = w["W"]
    n = min(4096, w["B"])
    from diffco_amd import _ops
    X = _ops.fkine(w["desc"], w["q"][:n]).reshape(n, -1).cpu()

    def run():
        x = X.clone().requires_grad_(True)
        if w["kspec"][0] == 0:
            kv = 1 / (1 + w["kspec"][1] / w["kspec"][2] * torch.cdist(x, sup).square()) ** w["kspec"][2]
        else:
            kv = torch.cdist(x, sup) / w["kspec"][2]
        (kv @ Wt).sum().backward()
        return x.grad
    run()
    best, t_start, reps = 1e30, time.perf_counter(), 0
    while reps < 5 and time.perf_counter() - t_start < budget_s:
        t0 = time.perf_counter()
        run()
        best = min(best, time.perf_counter() - t0)
        reps += 1
    return {"value": round(n / best / 1e6, 5), "unit": "M evals/s", "cores": nthr,
            "sample": f"{n} configs, torch {torch.__version__} CPU cdist+matmul+backward on precomputed features (no FK)"}


def load_pmc_traffic(name):
    """HBM bytes per launch from a committed rocprofv3 --pmc summary (profiles/pmc_<workload>.json), if any"""
    p = os.path.join(ROOT, "profiles", f"pmc_{name}.json")
    if os.path.exists(p):
        try:
            with open(p) as f:
                return json.load(f).get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gather of scores")
    ap.add_argument("--force-dist", action="store_true",
                    help="exercise the N>1 code path (process group + overlapped all-gather) even with one rank")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL writes its version banner to stdout
    # when the first communicator comes up), so file descriptor 1 is pointed at stderr for the whole run and the JSON
    # line goes to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                     "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    from diffco_amd import _lib
    _lib.require_gpu()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    w = make_workload(args.workload, args.batch, dev, seed=rank)
    m, q, B, C, dof = w["model"], w["q"], w["B"], w["C"], w["dof"]
    import ctypes as Ct
    lib = _lib.load()
    score = torch.empty((B, C), device=dev, dtype=torch.float32)
    grad = torch.empty((B, dof), device=dev, dtype=torch.float32)
    # N > 1: scores are all-gathered in BUCKETS of GATHER_EVERY steps (fewer, larger collectives: xGMI rings are
    # latency-bound at 256 KB per rank, and each call costs ~30 us of host time against a 118 us step), two buckets in
    # flight so that a bucket's gather overlaps the next bucket's sweeps
    K = GATHER_EVERY
    gathered = [torch.empty((world * K * B, C), device=dev, dtype=torch.float32) for _ in range(2)] if multi else None
    bucket = [torch.empty((K * B, C), device=dev, dtype=torch.float32) for _ in range(2)] if multi else None
    comm_stream = torch.cuda.Stream(dev) if multi else None
    qp, gp = Ct.c_void_p(q.data_ptr()), Ct.c_void_p(grad.data_ptr())

    traj = None
    if w["name"] == "cfg5":
        traj = traj_state(w, B // 50, 50, dev)

    def step(i, pending):
        if traj is not None:
            st = Ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.dcx_traj_adam_run(m._h, Ct.byref(traj[0]), Ct.byref(traj[1]), i + 1, 1, st))
            return
        b = (i // K) & 1
        out = bucket[b][(i % K) * B:(i % K + 1) * B] if multi else score
        if multi and i % K == 0 and pending[b] is not None:
            pending[b].wait()  # this bucket's previous gather (two buckets ago) must be done before it is rewritten
            pending[b] = None
        # ONE launch of the hot path on torch's current stream
        st = Ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.dcx_score_grad(m._h, qp, B, None, Ct.c_void_p(out.data_ptr()), gp, st))
        if multi and not args.no_gather and (i % K == K - 1 or i == last_step[0]):
            # all-gather of this bucket's scores on a side stream, overlapped with the next bucket's sweeps
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(ev)
                pending[b] = dist.all_gather_into_tensor(gathered[b], bucket[b], async_op=True)

    def drain(pending):
        for h in pending:
            if h is not None:
                h.wait()
        if comm_stream is not None:
            torch.cuda.current_stream(dev).wait_stream(comm_stream)

    last_step = [args.warmup - 1]  # a partly filled last bucket is gathered too
    pending = [None, None]
    for i in range(args.warmup):
        step(i, pending)
    drain(pending)
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pending = [None, None]
    last_step[0] = args.steps - 1
    t0 = time.perf_counter()
    e0.record()
    for i in range(args.steps):
        step(i, pending)
    e1.record()      # closes the sweep kernels on the launch stream (HIP events, same stream as the launches)
    drain(pending)
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    kern_ms = e0.elapsed_time(e1) / args.steps  # average launch-to-launch duration of the sweep kernel

    if multi:
        tt = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())
        tk = torch.tensor([kern_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tk, op=dist.ReduceOp.MAX)
        kern_ms = float(tk.item())

    if rank == 0:
        evals = world * B * args.steps
        value = evals / wall / 1e6
        F = flops_per_eval(w["D"], C, w["S"])
        ach_tf = F * B / (kern_ms * 1e-3) / 1e12
        ach_gbs = bytes_per_eval(dof, C) * B / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "million collision-score+grad evals/sec, 7-DoF FK-kernel, 2k supports",
            "value": round(value, 3), "unit": "M evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{w['name']}: {w['text']}", "batch_per_gpu": B, "global_batch": world * B,
                       "supports": w["S"], "features": w["D"], "classes": C,
                       "parallelism": f"batch-sharded x{world}, model replicated" +
                                      ("" if world == 1 else (", no gather" if args.no_gather else
                                                              f", RCCL all-gather of scores every {GATHER_EVERY} steps, overlapped")),
                       "launches_per_step": 2 if traj is not None else 1},
            "roofline": {"bound": "valu", "achieved": round(ach_tf, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach_tf / PEAK_FP32_TFLOPS, 4), "traffic": load_pmc_traffic(w["name"]),
                         "kernel": "dcx::score_kernel<D,KF,C,MODE>", "kernel_ms": round(kern_ms, 5),
                         "flops_per_eval": F,
                         "note": "fp32 VALU bound (peak == fp32 MFMA peak 157.3 TFLOP/s); algorithmic flops "
                                 "S*(5D+4C+6)+800 per eval (SURVEY.md §8d)",
                         "hbm": {"achieved": round(ach_gbs, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": round(ach_gbs / PEAK_HBM_GBS, 5),
                                 "bytes_per_eval": bytes_per_eval(dof, C)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
            tb = torch_cpu_baseline(w)
            if tb:
                out["cpu_baseline_torch"] = tb
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
