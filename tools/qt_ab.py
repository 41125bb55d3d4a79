#!/usr/bin/env python3
"""tools/qt_ab.py — the 16-configuration tile (score_kernel.h QT) against the split launch, same box, interleaved.
   python tools/qt_ab.py            (microseconds per score + gradient kernel: bench.py's measure(), HIP events on the launch stream, best of 3 x 2)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from diffco_amd import _lib  # noqa: E402

lib = _lib.require_gpu()
dev = torch.device("cuda", 0)


def timed(w, n=300):
    """the bench's own loop: HIP events around every launch on the launch stream, microseconds per kernel"""
    loop = bench.ScoreLoop(w, dev, 1, "none")
    return min(bench.measure(loop, n, 20, dev, False)[1] for _ in range(3)) * 1e3


cases = [("cfg2", b) for b in (256, 512, 1024, 2048, 3072, 4096)] + [("headline", b) for b in (1024, 2048, 4096)] + \
        [("headline_rq", 4096)]
if len(sys.argv) > 1:
    cases = [(sys.argv[1], int(b)) for b in sys.argv[2:]]
for name, B in cases:
    try:
        w = bench.make_workload(name, B, dev)
    except Exception as exc:  # noqa: BLE001
        print(name, B, "skipped:", exc)
        continue
    m, q = w["model"], w["q"]
    res = {}
    for rnd in range(2):
        for v in (0, 1):
            lib.dcx_debug_set(b"qt", v)
            res.setdefault(v, []).append(timed(w))
    lib.dcx_debug_set(b"qt", -1)
    s0, g0 = m.score_grad_raw(q)
    lib.dcx_debug_set(b"qt", 0)
    s1, g1 = m.score_grad_raw(q)
    lib.dcx_debug_set(b"qt", -1)
    rel = float((g0 - g1).abs().max() / g1.abs().max())
    a, b = min(res[0]), min(res[1])
    print(f"{name:12s} B={B:5d} S={w['S']:5d} D={w['D']:3d}  split launch {a:7.2f} us   16-configuration tile {b:7.2f} us   {100 * (b / a - 1):+6.1f} %   "
          f"(rule: {timed(w, 100):7.2f} us; gradients differ by {rel:.1e})", flush=True)
