#!/usr/bin/env python3
"""tools/hess_probe.py — developer tool (GPU box): time of dcx_score_hess against the route it replaced (central
differences of the analytic gradient: 2 * dof probes per point through dcx_score_grad) and both routes' error against
central differences of the float64 oracle's gradient, at trust-constr-sized batches of the Baxter config-#2 model."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle  # noqa: E402

dev = torch.device("cuda", 0)
for name, B in (("cfg2", 256), ("cfg2", 2048), ("headline", 2048), ("cfg3", 512)):
    w = bench.make_workload(name, B, dev)
    m, q = w["model"], w["q"]
    dof = q.shape[1]
    up = torch.randn((B, w["C"]), device=dev) if w["C"] > 1 else None

    def timed(fn, n=50):
        # keep the GPU busy for ~100 ms first: its clocks ramp for ~50 ms after idling (tools/clock_ramp.py)
        import time
        t_end = time.perf_counter() + 0.1
        while time.perf_counter() < t_end:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(n):
            out = fn()
        t1.record()
        torch.cuda.synchronize()
        return t0.elapsed_time(t1) / n * 1e3, out

    eps = 4e-3
    eye = torch.eye(dof, device=dev)
    probes = (q[:, None, None, :] + eps * torch.stack([eye, -eye])[None]).reshape(-1, dof).contiguous()
    upp = None if up is None else up[:, None, None, :].expand(B, 2, dof, w["C"]).reshape(-1, w["C"]).contiguous()
    t_fd, (_, gp) = timed(lambda: m.score_grad_raw(probes, upp))
    gp = gp.reshape(B, 2, dof, dof)
    H_fd = ((gp[:, 0] - gp[:, 1]) / (2 * eps)).double().cpu().numpy()
    t_an, (_, H) = timed(lambda: m.score_hess_raw(q, up))
    H_an = H.double().cpu().numpy()
    n64 = 32
    q64 = w["q_cpu"][:n64].numpy().astype(np.float64)
    up64 = None if up is None else up[:n64].cpu().numpy().astype(np.float64)
    Ho = np.empty((n64, dof, dof))
    for i in range(dof):
        e = np.zeros(dof)
        e[i] = 1e-5
        _, gpl, _ = oracle.score_grad(w["desc"], *w["kspec"], w["sup"].cpu().numpy(), w["W"].numpy(), q64 + e, upstream=up64, dtype=np.float64)
        _, gmi, _ = oracle.score_grad(w["desc"], *w["kspec"], w["sup"].cpu().numpy(), w["W"].numpy(), q64 - e, upstream=up64, dtype=np.float64)
        Ho[:, i, :] = (gpl - gmi) / 2e-5
    rel = lambda a: float(np.abs(a[:n64] - Ho).max() / np.abs(Ho).max())  # noqa: E731
    print(f"{name:<9} points={B:<5} S={w['S']:<5} C={w['C']}  analytic {t_an:8.1f} us (err {rel(H_an):.1e})   "
          f"gradient differences {t_fd:8.1f} us (err {rel(H_fd):.1e})", flush=True)
    if (name, B) == ("cfg2", 256):  # the same launch with 16 supports: what is left is launch + the dual FK walks + hand-over
        from diffco_amd import _ops
        m16 = _ops.ScoreModel(w["desc"], *w["kspec"], w["sup"][:16].contiguous(), w["W"][:16].to(dev).contiguous(), device=dev)
        t16, _ = timed(lambda: m16.score_hess_raw(q, up))
        tg16, _ = timed(lambda: m16.score_grad_raw(q, up))
        print(f"          the same with S = 16: analytic {t16:8.1f} us   one gradient launch {tg16:8.1f} us", flush=True)
