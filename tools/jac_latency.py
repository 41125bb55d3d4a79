#!/usr/bin/env python3
"""tools/jac_latency.py — developer tool (GPU box): time of `dcx_score_jac` (all C Jacobian rows) for small batches of
BASELINE config #3's model (C = 5), as ONE launch with the classes in grid z vs one launch per class
(dcx_debug_set("jac_per_class", 1)).  HIP-event time over back-to-back calls."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
for B in (64, 256, 1024, 4096, 8192):
    w = bench.make_workload("cfg3", B, dev)
    m, q = w["model"], w["q"]
    row = []
    for per_class in (False, True):
        m._lib.dcx_debug_set(b"jac_per_class", 1 if per_class else -1)
        for _ in range(5):
            s, j = m.score_jac_raw(q)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(50):
            s, j = m.score_jac_raw(q)
        t1.record()
        torch.cuda.synchronize()
        row.append((t0.elapsed_time(t1) / 50 * 1e3, s.clone(), j.clone()))
    diff = float((row[0][2] - row[1][2]).abs().max() / row[1][2].abs().max())  # the two routes may pick different splits
    print(f"B={B:<6} one launch {row[0][0]:8.1f} us   per class {row[1][0]:8.1f} us   max rel. difference {diff:.1e}")
