#!/usr/bin/env python3
"""developer probe: score-only launches (dcx_score: MODE_SCORE sweeps, 11 VALU per pair in the expanded form) under the wave groups'
slice shares - DCX_SKEW picks them (0 = equal)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda", 0)
for wl, B in (("headline", 65536), ("headline", 16384), ("headline", 8192), ("cfg3", 65536), ("cfg3", 8192), ("cfg2", 4096)):
    w = bench.make_workload(wl, B, dev)
    m, q = w["model"], w["q"]
    for _ in range(300): m.score_raw(q)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 400
    for _ in range(2000 if B < 65536 else 1500): m.score_raw(q)
    e0.record()
    for _ in range(n): m.score_raw(q)
    e1.record(); torch.cuda.synchronize()
    print(f"DCX_SKEW={os.environ.get('DCX_SKEW')} DCX_SKEW8={os.environ.get('DCX_SKEW8')} {wl} B={B}: score-only {e0.elapsed_time(e1) / n * 1e3:.2f} us")
