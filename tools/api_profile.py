#!/usr/bin/env python3
"""tools/api_profile.py — developer tool (GPU box): where the host time of one `poly_score` forward + backward goes
(cProfile, cumulative, B = 50 Baxter waypoints against 2000 supports: what optim.adam_traj_optimize calls per iteration)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffco_amd import kernel, model  # noqa: E402
from diffco_amd.kernel_perceptrons import DiffCo  # noqa: E402

rob = model.BaxterLeftArmFK()
lim = rob.limits
S, B = 2000, int(sys.argv[1]) if len(sys.argv) > 1 else 50
torch.manual_seed(0)
dev = torch.device("cuda")
dc = DiffCo(kernel_func=kernel.RQKernel(10.0), transform=rob.fkine)
sq = torch.rand(S, 7) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
dc.support_points = sq.to(dev)
dc.support_transformed = rob.fkine(sq.to(dev))
dc.gains = torch.randn(S, device=dev)
dc.rbf_kernel = kernel.Polyharmonic(1, 1.0)
dc.rbf_nodes = torch.randn(S, device=dev)
q = (torch.rand(B, 7) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).to(dev).requires_grad_(True)


def fwd_bwd():
    s = dc.poly_score(q)
    (g,) = torch.autograd.grad(s.sum(), q)
    return g


for _ in range(50):
    fwd_bwd()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    fwd_bwd()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
