#!/usr/bin/env python3
"""developer probe: dcx_score_jac (C one-hot sweeps in one launch) and dcx_score_hess under the wave groups' slice shares (DCX_SKEW)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda", 0)
def t(fn, n=200, settle=600):
    for _ in range(settle): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for wl, B in (("cfg3", 65536), ("cfg3", 8192), ("cfg3", 1024), ("cfg3", 256), ("headline", 65536), ("headline", 8192), ("headline", 1024), ("headline", 300), ("headline", 128), ("headline", 50)):
    w = bench.make_workload(wl, B, dev)
    m, q = w["model"], w["q"]
    up = torch.ones((B, w["C"]), device=dev)
    print(f"DCX_SKEW={os.environ.get('DCX_SKEW')} {wl} B={B}: jac {t(lambda: m.score_jac_raw(q)):.2f} us   hess {t(lambda: m.score_hess_raw(q, up), 100, 200):.2f} us", flush=True)
