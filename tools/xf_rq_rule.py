#!/usr/bin/env python3
"""tools/xf_rq_rule.py — developer tool (GPU box): how the error of the EXPANDED sweep for RQKernel(p = 2) grows with
gamma * max |s - c|^2 (c = support centroid), against the direct form, both measured against the float64 CPU oracle on 2048
configurations.  The data behind dcx_model_create's rule for taking the expanded form with RQ (csrc/dcx_api.hip xf_rq_ok)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffco_amd import _lib, _ops, model  # noqa: E402
from oracle import oracle  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.require_gpu()
g = torch.Generator().manual_seed(0)
for rob_name, rob in (("baxter", model.BaxterLeftArmFK()), ("panda", model.PandaFK())):
    lo, hi = rob.limits[:, 0], rob.limits[:, 1]
    desc = rob.fk_desc()
    S, B = 2000, 2048
    sup_q = torch.rand((S, len(lo)), generator=g) * (hi - lo) + lo
    q = (torch.rand((B, len(lo)), generator=g) * (hi - lo) + lo).to(dev)
    sup = _ops.fkine(desc, sup_q.to(dev)).reshape(S, -1)
    ss_max = float(((sup - sup.mean(0)) ** 2).sum(1).max())
    for C in (1, 5):
        W = torch.randn((S, C), generator=g)
        for gamma in (2.0, 5.0, 10.0, 20.0, 40.0, 80.0):
            m = _ops.ScoreModel(desc, 0, gamma, 2.0, sup, W.to(dev), device=dev)
            so, go, _ = oracle.score_grad(desc, 0, gamma, 2.0, sup.cpu().numpy().astype(np.float64), W.numpy().astype(np.float64),
                                          q.cpu().numpy().astype(np.float64), dtype=np.float64)
            out = []
            for mode in (0, 2):   # 2 = the expanded form even where the rule says no
                lib.dcx_debug_set(b"xf", mode)
                s, gr = m.score_grad_raw(q)
                torch.cuda.synchronize()
                es = float(np.abs(s.cpu().numpy().astype(np.float64).reshape(so.shape) - so).max() / np.abs(so).max())
                eg = float(np.abs(gr.cpu().numpy().astype(np.float64) - go).max() / np.abs(go).max())
                out.append((es, eg))
            lib.dcx_debug_set(b"xf", -1)
            print(f"{rob_name:<7} C={C} gamma={gamma:5.1f}  gamma*max|s-c|^2 = {gamma * ss_max:7.1f}   direct {out[0][0]:.1e} / {out[0][1]:.1e}   "
                  f"expanded {out[1][0]:.1e} / {out[1][1]:.1e}", flush=True)
