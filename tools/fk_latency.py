#!/usr/bin/env python3
"""tools/fk_latency.py — developer tool (GPU box): single-wave latency of the FK forward and of forward + J^T, i.e. the
prologue / epilogue of one block of the fused kernel, from one-block launches (B = 64) of dcx_fkine / dcx_fkine_vjp.
Run under `rocprofv3 --kernel-trace --stats` and read the average kernel durations."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from diffco_amd import model  # noqa: E402

for name, rob in (("baxter", model.BaxterLeftArmFK()), ("panda", model.PandaFK()), ("urdf_panda", H.urdf_robot("urdf_panda"))):
    q = (torch.rand(64, rob.dof, device="cuda") - 0.5).requires_grad_(True)
    for _ in range(60):
        X = rob.fkine(q)
        (g,) = torch.autograd.grad(X.sum(), q)
    torch.cuda.synchronize()
    print(name, "done")
