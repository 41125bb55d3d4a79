#!/usr/bin/env python3
"""tools/two_stream_probe.py — what do the ~10 us a B = 65536 launch spends outside its steady-state sweep (first FK chains, last fold +
J^T, clock ramp) cost a caller that has INDEPENDENT batches to score?  The same K launches issued on one stream
(bench.py's loop) and alternating over two / three streams, so that launch i + 1's first blocks start while launch i's last ones
drain.  Results are per launch (total time / K); every launch writes its own output buffers."""
import ctypes as Ct
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from diffco_amd import _lib  # noqa: E402

lib = _lib.require_gpu()
dev = torch.device("cuda", 0)
for wl, B in (("headline", 65536), ("headline", 16384), ("cfg3", 65536), ("cfg4", 1 << 20)):
    w = bench.make_workload(wl, B, dev)
    m, q = w["model"], w["q"]
    K = 400 if B <= 65536 else 24
    res = {}
    for ns in (1, 2, 3):
        streams = [torch.cuda.Stream(dev) for _ in range(ns)]
        outs = [(torch.empty((B, w["C"]), device=dev), torch.empty((B, w["dof"]), device=dev)) for _ in range(ns)]

        def run(n):
            for i in range(n):
                s = streams[i % ns]
                o, g = outs[i % ns]
                _lib.check(lib.dcx_score_grad(m._h, Ct.c_void_p(q.data_ptr()), B, None, Ct.c_void_p(o.data_ptr()), Ct.c_void_p(g.data_ptr()),
                                              Ct.c_void_p(s.cuda_stream)))
        run(K)           # settle
        torch.cuda.synchronize()
        run(K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(K)
        torch.cuda.synchronize()
        res[ns] = (time.perf_counter() - t0) / K * 1e6
    ref = outs[0][1].clone()
    print(f"{wl:9s} B={B:8d}: one stream {res[1]:8.2f} us per launch   two streams {res[2]:8.2f}   three {res[3]:8.2f}   "
          f"(x{res[1] / res[2]:.3f}, x{res[1] / res[3]:.3f})")
