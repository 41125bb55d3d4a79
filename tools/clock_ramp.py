#!/usr/bin/env python3
"""tools/clock_ramp.py — developer tool (GPU box): time per launch of the headline sweep as a function of how long the GPU
has been busy (chunks of 100 launches, HIP events), after an idle period: how many untimed launches bench.py has to issue
before its warm-up for the clocks to be where a production loop runs them."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "headline"
w = bench.make_workload(name, bench.WORKLOADS[name][4], dev)
m, q = w["model"], w["q"]
for idle in (0.0, 1.0):
    torch.cuda.synchronize()
    time.sleep(idle)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    evs[0].record()
    for c in range(60):
        for _ in range(100):
            m.score_grad_raw(q)
        evs[c + 1].record()
    torch.cuda.synchronize()
    us = [evs[c].elapsed_time(evs[c + 1]) * 10 for c in range(60)]
    t = 0.0
    out = []
    for c, u in enumerate(us):
        t += u * 100 / 1e3
        if c < 10 or c % 5 == 4:
            out.append(f"{t:7.1f} ms: {u:6.1f}")
    print(f"{name}, after {idle:.0f} s idle — elapsed busy time: us per launch (chunks of 100 launches)")
    print("   ".join(out))
