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
def offsets():
    """`tools/xf_probe.py offsets`: the headline shape with the Baxter arm's base translated — the expanded form's time and
    error must not depend on where the scene stands (the rows and the features are shifted by the support centroid)"""
    from diffco_amd import _fkdesc, _ops, model
    rob = model.BaxterLeftArmFK()
    lim = rob.limits.float()
    g = torch.Generator().manual_seed(0)
    S, B = 2000, 65536
    rnd = lambda n: torch.rand((n, 7), generator=g) * (lim[:, 1] - lim[:, 0]) + lim[:, 0]  # noqa: E731
    sup_q, q = rnd(S), rnd(B).to(dev)
    W = torch.randn((S, 1), generator=g)
    for where in ((0.0, 0.0, 0.0), (20.0, 0.0, 0.0), (100.0, 50.0, 0.0), (1000.0, -500.0, 30.0)):
        base = list(_fkdesc.IDENTITY_BASE)
        base[3], base[7], base[11] = where
        desc = _fkdesc.dh_desc(7, [rob.dhparams.chain(range(7), base)], [(0, i, (0, 0, 0)) for i, mk in enumerate(rob.fk_mask) if mk])
        sup = _ops.fkine(desc, sup_q.to(dev)).reshape(S, -1)
        m = _ops.ScoreModel(desc, 1, 1.0, 1.0, sup, W.to(dev), device=dev)
        so, go, _ = oracle.score_grad(desc, 1, 1.0, 1.0, sup.cpu().numpy().astype(np.float64), W.numpy().astype(np.float64),
                                      q[:2048].cpu().numpy().astype(np.float64), dtype=np.float64)
        out = []
        for mode in (0, 1):
            lib.dcx_debug_set(b"xf", mode)
            best = 1e30
            for _ in range(3):
                for _ in range(5):
                    s, gr = m.score_grad_raw(q)
                torch.cuda.synchronize()
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
                for _ in range(50):
                    s, gr = m.score_grad_raw(q)
                t1.record()
                torch.cuda.synchronize()
                best = min(best, t0.elapsed_time(t1) / 50 * 1e3)
            es = float(np.abs(s[:2048].cpu().numpy().astype(np.float64).reshape(so.shape) - so).max() / np.abs(so).max())
            eg = float(np.abs(gr[:2048].cpu().numpy().astype(np.float64) - go).max() / np.abs(go).max())
            out.append((best, es, eg))
        lib.dcx_debug_set(b"xf", -1)
        print(f"base at {where}: direct {out[0][0]:7.1f} us (err {out[0][1]:.1e} / {out[0][2]:.1e})   "
              f"expanded {out[1][0]:7.1f} us (err {out[1][1]:.1e} / {out[1][2]:.1e})", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "offsets":
    offsets()
    sys.exit(0)
if len(sys.argv) > 1:
    cases = [(a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:]]
for name, B in cases:
    w = bench.make_workload(name, B, dev)
    m, q = w["model"], w["q"]
    up = torch.randn((B, w["C"]), device=dev) if w["C"] > 1 else None
    res = {}
    for mode in (0, 1, 0, 1):
        lib.dcx_debug_set(b"xf", 2 if (mode and os.environ.get("XF_FORCE")) else mode)   # XF_FORCE=1: past the RQ rule (dcx_api.hip xf_rq_ok)
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
