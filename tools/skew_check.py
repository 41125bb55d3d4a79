#!/usr/bin/env python3
"""developer check: score / gradient with the wave groups' skewed slices against equal slices, many support counts and batches"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diffco_amd import _lib, _ops, model
lib = _lib.require_gpu()
rob = model.BaxterLeftArmFK()
lim = rob.limits
g = torch.Generator().manual_seed(1)
def setk(n, v): _lib.check(lib.dcx_debug_set(n.encode(), v))
bad = 0
for qt in (0, -1):
    setk("qt", qt)
    for kind, p0, p1 in ((1, 1.0, 1.0), (0, 10.0, 2.0)):
        for S in (64, 100, 200, 300, 301, 500, 1000, 2000):
            sq = torch.rand((S, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
            sup = _ops.fkine(rob.fk_desc(), sq.cuda()).reshape(S, -1)
            m = _ops.ScoreModel(rob.fk_desc(), kind, p0, p1, sup, (0.05 * torch.randn(S, generator=g)).cuda())
            for B in (1, 20, 35, 64, 200, 5000):
                q = (torch.rand((B, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]).cuda()
                setk("skew", 0); s0, g0 = m.score_grad_raw(q)
                srt = s0.reshape(-1).sort().values
                mg = float(0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])) if len(srt) > 1 else float(srt[0]) - 1.0
                sh0, gh0 = m.score_hinge_grad_raw(q, mg, -1.0)
                setk("skew", -1); s1, g1 = m.score_grad_raw(q); sh1, gh1 = m.score_hinge_grad_raw(q, mg, -1.0)
                e = [float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in ((s1, s0), (g1, g0), (gh1, gh0))]
                if max(e) > 1e-5:
                    bad += 1
                    print(f"qt={qt} kind={kind} S={S} B={B}: score {e[0]:.2e} grad {e[1]:.2e} hinge-grad {e[2]:.2e}")
print("mismatches:", bad)
