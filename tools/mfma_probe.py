#!/usr/bin/env python3
"""tools/mfma_probe.py — developer tool (GPU box): headline / config shapes with the gradient fold on the VALU vs on the
matrix cores (dcx_debug_set("mfma", 0/1)): max relative difference and HIP-event time per launch."""
import ctypes as Ct
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from diffco_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.require_gpu()
for name, B in (("headline", 65536), ("headline", 1 << 20), ("headline", 4096), ("cfg2", 4096), ("cfg3", 8192), ("cfg3", 65536),
                ("cfg4", 1 << 18), ("cfg5", 12800)):
    w = bench.make_workload(name, B, dev)
    m, q = w["model"], w["q"]
    res = {}
    for mode in (0, 1, 0, 1):
        lib.dcx_debug_set(b"mfma", mode)
        for _ in range(5):
            s, g = m.score_grad_raw(q)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50 if B <= 65536 else 10
        t0.record()
        for _ in range(n):
            s, g = m.score_grad_raw(q)
        t1.record()
        torch.cuda.synchronize()
        res.setdefault(mode, []).append((t0.elapsed_time(t1) / n * 1e3, s.clone(), g.clone()))
    a, b = res[0][-1], res[1][-1]
    ds = float((a[1] - b[1]).abs().max() / a[1].abs().max())
    dg = float((a[2] - b[2]).abs().max() / a[2].abs().max())
    F = bench.flops_per_eval(w["D"], w["C"], w["S"]) * B
    print(f"{name:<9} B={B:<8} valu {min(r[0] for r in res[0]):9.1f} us ({F / min(r[0] for r in res[0]) / 1e6 / 157.3:.3f})   "
          f"mfma {min(r[0] for r in res[1]):9.1f} us ({F / min(r[0] for r in res[1]) / 1e6 / 157.3:.3f})   "
          f"rel diff score {ds:.1e} grad {dg:.1e}", flush=True)
