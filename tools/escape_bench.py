#!/usr/bin/env python3
"""What the escape loop (diffco_amd.escape.OptimSampler, reference scripts/escape.py:19-38) costs on an MI355X, in the call
patterns of the reference's scripts, beside the same loop on torch-CPU expressions of the reference's score.

  one configuration at a time (scripts/compare_sampling.py:177-195: N_WAYPOINTS 3, lr 0.2, wrap2pi, last configuration only)
      fused (dcx_escape_adam, one read-back) / host loop through autograd on the HIP score / torch-CPU
  B configurations together (optim_escape_batch): escapes per second for B = 64 ... 65536, N_WAYPOINTS 20

Usage: python tools/escape_bench.py            (on the GPU box; prints a table, kept as profiles/r05_escape.txt)
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timed(fn, reps, sync=True):
    fn()
    if sync:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    if sync:
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    from diffco_amd import kernel, model, utils
    from diffco_amd.escape import OptimSampler
    from diffco_amd.kernel_perceptrons import DiffCo
    torch.manual_seed(0)
    rob = model.BaxterLeftArmFK()
    lim = rob.limits
    S = 2000
    sup_q = torch.rand(S, 7) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    w = torch.randn(S) * 0.02 + 0.001
    dc = DiffCo(transform=rob.fkine)
    dc.support_points, dc.support_transformed = sup_q.cuda(), rob.fkine(sup_q.cuda())
    dc.rbf_kernel, dc.rbf_nodes = kernel.Polyharmonic(1, 1.0), w.cuda()
    starts = (torch.rand(65536, 7) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).cuda()
    s0 = dc.poly_score(starts[:4096])
    margin = float(s0.median()) - 0.2
    print(f"Baxter left arm, Polyharmonic(1, 1) spline score, S = {S}; margin = median score - 0.2")

    # ---- one configuration at a time ---------------------------------------------------------------------------------
    opts = {"N_WAYPOINTS": 3, "safety_margin": margin, "lr": 0.2, "record_freq": None, "post_transform": utils.wrap2pi}
    fused = OptimSampler(rob, dc.poly_score, opts)
    host = OptimSampler(rob, dc.poly_score, dict(opts, post_transform=lambda x: utils.wrap2pi(x)))
    one = starts[int(torch.argmax(s0))][None].clone()      # deep in collision: all three steps are taken
    n_f = fused.optim_escape(one)[1]
    n_h = host.optim_escape(one)[1]
    t_f = timed(lambda: fused.optim_escape(one), 300, sync=False)
    t_h = timed(lambda: host.optim_escape(one), 100, sync=False)
    print(f"one configuration, N_WAYPOINTS 3 ({n_f} / {n_h} evaluations), start on the device:")
    print(f"    fused (one library call, one read-back)   {t_f * 1e6:8.1f} us per escape")
    print(f"    host loop, autograd on the HIP score      {t_h * 1e6:8.1f} us per escape")
    one_c = one.cpu()
    t_fc = timed(lambda: fused.optim_escape(one_c), 300, sync=False)
    print(f"    fused, start and result on the host       {t_fc * 1e6:8.1f} us per escape")

    # the same loop on torch-CPU expressions of the reference's score (what the reference itself runs on the host cores)
    from helpers import TorchDHRobot, TorchKernel
    rob_t = TorchDHRobot(rob)
    sup_t = rob_t.fkine(sup_q.double()).reshape(S, -1).float()
    w_t, kern_t = w.clone(), TorchKernel("poly1", 1, 1.0)

    def cpu_score(p):
        return kern_t(rob_t.fkine(p).reshape(-1, sup_t.shape[1]), sup_t) @ w_t
    cpu = OptimSampler(rob_t, cpu_score, opts)
    n_c = cpu.optim_escape(one_c)[1]
    t_c = timed(lambda: cpu.optim_escape(one_c), 20, sync=False)
    print(f"    torch-CPU expressions of the same loop    {t_c * 1e6:8.1f} us per escape ({n_c} evaluations, {torch.get_num_threads()} torch threads)")

    # ---- B independent loops together ----------------------------------------------------------------------------------
    batch = OptimSampler(rob, dc.poly_score, {"N_WAYPOINTS": 20, "safety_margin": margin, "lr": 5e-2, "record_freq": None,
                                              "post_transform": utils.wrap2pi})
    print("B independent loops, N_WAYPOINTS 20, lr 0.05; compact_every = 0: every step sweeps all B rows (stopped loops are left alone),")
    print("k > 0: the loops that stopped are taken out of the sweep after every k-th step (one stream synchronisation each):")
    for B in (64, 1024, 4096, 16384, 65536, 262144):
        q = (starts[:B] if B <= len(starts) else starts.repeat((B + len(starts) - 1) // len(starts), 1)[:B] + 1e-3 * torch.randn(B, 7, device="cuda")).contiguous()
        line = f"    B = {B:6d}"
        for k in (0, 1, 2, 4):
            final, checks = batch.optim_escape_batch(q, compact_every=k)
            t = timed(lambda: batch.optim_escape_batch(q, compact_every=k), 20 if B <= 4096 else 5)
            line += f"   k={k}: {t * 1e3:7.3f} ms {B / t / 1e6:7.2f} M/s"
        free = float((checks < 20).float().mean())
        print(line + f"   ({float(checks.float().mean()):4.1f} evaluations per loop, {free * 100:4.1f} % free before step 20)")


if __name__ == "__main__":
    main()
