#!/usr/bin/env python3
"""tools/optim_wall.py — developer tool (GPU box): what a caller of the reference's Adam trajectory optimiser waits for, on the
golden problem (tests/golden/optim_adam_baxter: Baxter, 20 waypoints, 50 iterations, one trial, given init), three ways:
  (1) the drop-in `optim.adam_traj_optimize` with the HIP checker (`dc.poly_score`): the reference's Python loop, every
      dist_est(p) / robot.fkine(p) of it one HIP launch, torch autograd between them (CPU float64 path tensor, as the reference);
  (2) `fused_adam_traj_optimize`: the whole loop in one persistent launch;
  (3) the same Python loop on torch-CPU restatements of the robot and the checker (what the reference itself runs: its
      FK and kernel expressions in torch ops on the host cores).
Wall seconds per call (best of 3 after one warm-up call), iterations per second, and the records' agreement."""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import GOLDEN, TorchDHRobot, TorchKernel, load, make_robot, relerr  # noqa: E402
from diffco_amd import fused_adam_traj_optimize, kernel, optim  # noqa: E402
from diffco_amd.kernel_perceptrons import DiffCo  # noqa: E402

d = load("optim_adam_baxter")
options = json.load(open(os.path.join(GOLDEN, "optim_adam_baxter_options.json")))
options["init_solution"] = torch.from_numpy(d["init"]).clone()
start, target = torch.from_numpy(d["start"]), torch.from_numpy(d["target"])
rob = make_robot("baxter_left")
dc = DiffCo(transform=rob.fkine)
dc.support_points = torch.from_numpy(d["sup_q"])
dc.support_transformed = rob.fkine(dc.support_points)
dc.rbf_kernel, dc.rbf_nodes = kernel.Polyharmonic(1, 1.0), torch.from_numpy(d["weights"])
dc_gpu = DiffCo(transform=rob.fkine)
dc_gpu.support_points = dc.support_points.cuda()
dc_gpu.support_transformed = rob.fkine(dc_gpu.support_points)
dc_gpu.rbf_kernel, dc_gpu.rbf_nodes = kernel.Polyharmonic(1, 1.0), dc.rbf_nodes.cuda()

rob_t = TorchDHRobot(rob)
sup_t = rob_t.fkine(torch.from_numpy(d["sup_q"]).double()).reshape(len(d["sup_q"]), -1)
w_t = torch.from_numpy(d["weights"]).double()
kern_t = TorchKernel("poly1", 1, 1.0)


def dist_est_torch(p):
    return kern_t(rob_t.fkine(p).reshape(len(p), -1), sup_t) @ w_t[:, None]


def timed(fn, reps=3):
    with contextlib.redirect_stdout(io.StringIO()):
        rec = fn()
        best = 1e30
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rec = fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
    return best, rec


iters = options["MAXITER"]
print(f"golden problem: S = {len(d['sup_q'])} supports, W = {len(d['init'])} waypoints, MAXITER = {iters}, NUM_RE_TRIALS = {options['NUM_RE_TRIALS']}, "
      f"host cores = {os.cpu_count()}, torch threads = {torch.get_num_threads()}")
rows = [("drop-in adam_traj_optimize, HIP checker (CPU-resident supports)", lambda: optim.adam_traj_optimize(rob, dc.poly_score, start, target, dict(options))),
        ("drop-in adam_traj_optimize, HIP checker (GPU-resident supports)", lambda: optim.adam_traj_optimize(rob, dc_gpu.poly_score, start, target, dict(options))),
        ("fused_adam_traj_optimize (one persistent launch)", lambda: fused_adam_traj_optimize(rob, dc_gpu.poly_score, start, target, dict(options))),
        ("the same loop on torch-CPU robot + checker (reference expressions)", lambda: optim.adam_traj_optimize(rob_t, dist_est_torch, start, target, dict(options)))]
for name, fn in rows:
    wall, rec = timed(fn)
    n_it = rec["cnt_check"] / len(d["init"])
    print(f"{name:<70} {wall * 1e3:9.2f} ms per call   {wall / max(n_it, 1) * 1e6:9.1f} us per iteration   success {rec['success']}  "
          f"cost {rec['cost']:.5f} (reference {float(d['cost']):.5f})  solution vs reference {relerr(np.array(rec['solution']), d['solution']):.1e}")
