#!/usr/bin/env python3
"""tools/traj_spec_probe.py — the multi-class trajectory kernel's speculation (traj_fused.h): microseconds per Adam iteration of
config #5's loop on config #3's five-class model (256 restarts x 50 waypoints, S = 2000) for margins that make the hinge's
indicators flip often (0: half of the (waypoint, class) entries active, random restarts far from convergence), rarely (the 95 %
quantile of the scores) and never (above every score: free space).  DCX_LIB selects the library."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from diffco_amd.traj import ShardedAdamRun  # noqa: E402

dev = torch.device("cuda", 0)
for wl in ("cfg5_c5", "cfg5"):
    w = bench.make_workload(wl, 0, dev)
    m = w["model"]
    g = torch.Generator().manual_seed(4242)
    lo, hi = w["lo"], w["hi"]
    inits = torch.rand((256, bench.TRAJ_W, w["dof"]), generator=g) * (hi - lo) + lo
    s = m.score_raw(inits.reshape(-1, w["dof"]).to(dev))
    for label, margin in (("margin 0 (indicators flip)", 0.0), ("95 % quantile", float(s.quantile(0.95))), ("above every score", float(s.max()) + 100.0)):
        for lr in (0.05, 0.005):
            run = ShardedAdamRun(m, torch.stack([lo, hi], dim=1), inits, lr=lr, safety_margin=margin, max_speed=0.3, grad_tol=0.0)
            run.run(192)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                run.run(192)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (5 * 192) * 1e6
            print(f"{wl:8s} {label:28s} lr {lr:5.3f}: {dt:7.2f} us per iteration   (collision term of the last step: {float(run.t['stats'][:, 4].mean()):.3f})")
            run.close()
