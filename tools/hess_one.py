#!/usr/bin/env python3
"""developer probe: a few dcx_score_hess calls of one batch size (for rocprofv3 --kernel-trace --stats)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda", 0)
wl, B = sys.argv[1], int(sys.argv[2])
w = bench.make_workload(wl, B, dev)
m, q = w["model"], w["q"]
up = torch.ones((B, w["C"]), device=dev)
for _ in range(300): m.score_hess_raw(q, up)
torch.cuda.synchronize()
