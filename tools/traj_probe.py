#!/usr/bin/env python3
"""tools/traj_probe.py — developer tool (GPU box): BASELINE config #5 (R restarts x 50 waypoints, S = 2000) per-iteration
time of dcx_traj_adam_run, HIP events around ONE call of N iterations:
  two-launch loop (knob traj_fused = 0)  |  persistent launch, one workgroup per path (traj_ys = 1)  |  persistent launch,
  cluster form (traj_ys = rule or the values given): a path's supports split over ys workgroups (csrc/traj_fused.h).
    python tools/traj_probe.py [N] [R,R,...] [ys,ys,...]"""
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
Rs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256, 128, 64, 32, 16, 8]
yss = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [-1]
for R in Rs:
    w = bench.make_workload("cfg5", R * 50, dev)
    m, q, dof = w["model"], w["q"], w["dof"]
    modes = [("two-launch", 0, 1), ("one WG / path", 1, 1)] + [(f"cluster ys={'rule' if y < 0 else y}", 1, y) for y in yss]
    for rep in range(2):
        for label, fused, tys in modes:
            lib.dcx_debug_set(b"traj_fused", fused)
            lib.dcx_debug_set(b"traj_ys", tys)
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
            bad = float(bufs[6][:, 7].min())
            print(f"R={R:<4} {label:<18} {us:7.2f} us / iteration   {R * 50 / us:8.1f} M evals/s" + ("   EXCHANGE GAVE UP" if bad < 0 else ""), flush=True)
