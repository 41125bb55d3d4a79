#!/usr/bin/env python3
"""developer probe: score-only launches of batches beyond four tiles per CU (8-wave blocks) under DCX_SKEW8"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda", 0)
for wl, B in (("headline", 262144), ("headline", 1048576), ("cfg3", 262144), ("cfg2_panda", 262144)):
    w = bench.make_workload(wl, B, dev)
    m, q = w["model"], w["q"]
    n = 100 if B < 1000000 else 40
    for _ in range(4 * n): m.score_raw(q)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): m.score_raw(q)
    e1.record(); torch.cuda.synchronize()
    print(f"DCX_SKEW8={os.environ.get('DCX_SKEW8')} {wl} B={B}: score-only {e0.elapsed_time(e1) / n * 1e3:.1f} us", flush=True)
