#!/usr/bin/env python3
"""tools/bench_urdf.py — developer tool (GPU box): fused score+gradient throughput with URDF kinematic trees as the
transform (DCX_FK_TREE), next to the DH Panda of the same size.  Robots come from the joint tables stored with the
golden fixtures (tests/golden/fk_urdf_*.npz); supports and weights are synthetic (seeded)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from diffco_amd import _ops, model  # noqa: E402


def run(name, rob, S, B, iters=30):
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    lim = rob.limits.float()
    sq = torch.rand(S, rob.dof, generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
    q = (torch.rand(B, rob.dof, generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).to(dev)
    sup = rob.fkine(sq.to(dev)).reshape(S, -1)
    W = torch.randn(S, 1, generator=g).to(dev)
    m = _ops.ScoreModel(rob.fk_desc(), 1, 1.0, 1.0, sup, W)
    for _ in range(3):
        m.score_grad_raw(q)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        m.score_grad_raw(q)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    D = sup.shape[1]
    flops = S * (5 * D + 4 + 6) + 800
    print(f"{name:<22} dof={rob.dof:<3} D={D:<3} S={S} B={B:<7} {ms * 1e3:9.1f} us  {B / ms / 1e3:8.1f} M evals/s  "
          f"{B * flops / ms / 1e9:6.1f} TFLOP/s")


if __name__ == "__main__":
    S = 2000
    for B in (4096, 65536):
        run("panda DH (7 pts)", model.PandaFK(), S, B)
        for n in ("urdf_panda_nogripper", "urdf_panda", "urdf_fetch_arm", "urdf_jaco", "urdf_allegro", "urdf_fetch",
                  "urdf_iiwa7_allegro"):
            run(n, H.urdf_robot(n), S, B)
        run("urdf_dual_panda", H.dual_panda_robot(), S, B)
