#!/usr/bin/env python3
"""tools/traj_probe.py — developer tool (GPU box): BASELINE config #5 (256 restarts x 50 waypoints, S = 2000) per-iteration
time of dcx_traj_adam_run as the two-launch loop (knob traj_fused = 0) vs the persistent launch (1), HIP events around ONE
call of N iterations."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from diffco_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.require_gpu()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 192
for R in (256, 128, 32):
    w = bench.make_workload("cfg5", R * 50, dev)
    m, q, dof = w["model"], w["q"], w["dof"]
    for fused in (0, 1, 0, 1):
        lib.dcx_debug_set(b"traj_fused", fused)
        st, opt, bufs = bench.traj_state(w, R, 50, dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.dcx_traj_adam_run(m._h, C.byref(st), C.byref(opt), 1, 10, stream))
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        _lib.check(lib.dcx_traj_adam_run(m._h, C.byref(st), C.byref(opt), 11, N, stream))
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) / N * 1e3
        print(f"R={R:<4} fused={fused}  {us:7.2f} us / iteration   {R * 50 / us:8.1f} M evals/s", flush=True)
