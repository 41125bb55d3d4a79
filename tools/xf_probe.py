#!/usr/bin/env python3
"""tools/xf_probe.py — developer tool (GPU box, libdcx built with EXTRA=-DDCX_BOTH_FORMS): the sweep in its direct form
(differences, knob xf = 0) against the expanded form (score_kernel.h XF, knob xf = 1) on the bench workloads:
HIP-event time per launch, fraction of the fp32 peak, the two forms' relative difference, and each form's error
against the float64 CPU oracle on the first 2048 configurations."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from diffco_amd import _lib  # noqa: E402
from oracle import oracle  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.require_gpu()
cases = [("headline", 65536), ("headline", 1 << 20), ("headline", 4096), ("cfg2", 4096), ("cfg2_panda", 4096), ("cfg3", 8192),
         ("cfg3", 65536), ("cfg3_poly", 8192), ("cfg3_poly", 65536), ("cfg4", 1 << 18), ("cfg5", 12800)]
if len(sys.argv) > 1:
    cases = [(a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:]]
for name, B in cases:
    w = bench.make_workload(name, B, dev)
    m, q = w["model"], w["q"]
    up = torch.randn((B, w["C"]), device=dev) if w["C"] > 1 else None
    res = {}
    for mode in (0, 1, 0, 1):
        lib.dcx_debug_set(b"xf", mode)
        for _ in range(5):
            s, g = m.score_grad_raw(q, up)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50 if B <= 65536 else 10
        t0.record()
        for _ in range(n):
            s, g = m.score_grad_raw(q, up)
        t1.record()
        torch.cuda.synchronize()
        res.setdefault(mode, []).append((t0.elapsed_time(t1) / n * 1e3, s.clone(), g.clone()))
    lib.dcx_debug_set(b"xf", -1)
    a, b = res[0][-1], res[1][-1]
    ds = float((a[1] - b[1]).abs().max() / a[1].abs().max())
    dg = float((a[2] - b[2]).abs().max() / a[2].abs().max())
    n64 = min(B, 2048)
    so, go, _ = oracle.score_grad(w["desc"], *w["kspec"], w["sup"].cpu().numpy(), w["W"].numpy(), w["q_cpu"][:n64].numpy(),
                                  upstream=None if up is None else up[:n64].cpu().numpy(), dtype=np.float64)
    err = lambda t, r: float(np.abs(t.cpu().numpy().astype(np.float64).reshape(r.shape) - r).max() / np.abs(r).max())  # noqa: E731
    F = bench.flops_per_eval(w["D"], w["C"], w["S"]) * B
    td, tx = min(r[0] for r in res[0]), min(r[0] for r in res[1])
    print(f"{name:<10} B={B:<8} direct {td:9.1f} us ({F / td / 1e6 / 157.3:.3f})   expanded {tx:9.1f} us ({F / tx / 1e6 / 157.3:.3f})   "
          f"x{td / tx:.3f}   forms differ: score {ds:.1e} grad {dg:.1e}   vs fp64 oracle: direct {err(a[1][:n64], so):.1e}/{err(a[2][:n64], go):.1e}  "
          f"expanded {err(b[1][:n64], so):.1e}/{err(b[2][:n64], go):.1e}", flush=True)
