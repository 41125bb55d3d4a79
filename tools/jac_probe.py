#!/usr/bin/env python3
"""tools/jac_probe.py — developer tool (GPU box): the full Jacobian dcx_score_jac of config #3's model (C = 5) as C one-hot
sweeps (knob jac_one_sweep = 0) against ONE sweep over all classes (jac_kernel.h, knob 1): HIP-event time per call and the
two routes' difference.  RQ(10) and the Polyharmonic(1,1) nodes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from diffco_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.require_gpu()
for name in ("cfg3", "cfg3_poly"):
    for B in (65536, 8192, 1024, 256):
        w = bench.make_workload(name, B, dev)
        m, q = w["model"], w["q"]
        res = {}
        for mode in (0, 1, 0, 1):
            lib.dcx_debug_set(b"jac_one_sweep", mode if mode == 0 else (1 if B < 8192 else -1))
            for _ in range(3):
                s, j = m.score_jac_raw(q)
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            t0.record()
            for _ in range(n):
                s, j = m.score_jac_raw(q)
            t1.record()
            torch.cuda.synchronize()
            res.setdefault(mode, []).append((t0.elapsed_time(t1) / n * 1e3, s.clone(), j.clone()))
        lib.dcx_debug_set(b"jac_one_sweep", -1)
        a, b = res[0][-1], res[1][-1]
        dj = float((a[2] - b[2]).abs().max() / a[2].abs().max())
        print(f"{name:<10} B={B:<7} per-class route {min(r[0] for r in res[0]):9.1f} us   one sweep {min(r[0] for r in res[1]):9.1f} us   "
              f"x{min(r[0] for r in res[0]) / min(r[0] for r in res[1]):.2f}   jac differ {dj:.1e}", flush=True)
