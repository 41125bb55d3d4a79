#!/usr/bin/env python3
"""tools/soak_split.py — developer tool (GPU box): two minutes of split launches (the fence-free cross-block hand-over of
the fused gradient kernel and of dcx_score_hess) on four concurrent streams of one model (three with the arrival counters, the default stream with the owner-polls hand-over), every result compared bit for
bit with the first: config #2 / #3 shapes, ragged batches, Panda's wider rows."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
bad = 0
t0 = time.time()
for name, B in (("cfg2", 4096), ("cfg3", 8192), ("cfg2", 700), ("cfg2_panda", 2048), ("cfg3", 1500)):
    w = bench.make_workload(name, B, dev)
    m, q = w["model"], w["q"]
    up = torch.randn((B, w["C"]), device=dev) if w["C"] > 1 else None
    s0, g0 = m.score_grad_raw(q, up)
    s0, g0 = s0.clone(), g0.clone()
    H0 = m.score_hess_raw(q[:256], None if up is None else up[:256])[1].clone()
    n = 0
    # three side streams (arrival-counter hand-over) and the default stream (the ONE stream per device that may use the
    # owner-polls hand-over, round 5): both protocols side by side, bit-identical results expected from either
    streams = [torch.cuda.Stream() for _ in range(3)] + [torch.cuda.current_stream()]
    while time.time() - t0 < 25 * (1 + ["cfg2:4096","cfg3:8192","cfg2:700","cfg2_panda:2048","cfg3:1500"].index(f"{name}:{B}")):
        outs = []
        for it in range(200):
            st = streams[it % 4]
            with torch.cuda.stream(st):
                outs.append(m.score_grad_raw(q, up))
                if it % 50 == 0:
                    outs.append((None, m.score_hess_raw(q[:256], None if up is None else up[:256])[1]))
        torch.cuda.synchronize()
        for s, g in outs:
            if s is None:
                bad += int(not torch.equal(g, H0))
            else:
                bad += int(not (torch.equal(s, s0) and torch.equal(g, g0)))
        n += len(outs)
    print(name, B, "launches", n, "mismatches so far", bad, flush=True)
print("SOAK", "OK" if bad == 0 else "FAILED", bad)
