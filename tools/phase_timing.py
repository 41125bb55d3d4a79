#!/usr/bin/env python3
"""tools/phase_timing.py — developer tool: per-phase cycle stamps of block (0,0) of the fused kernel, from a
libdcx built with -DDCX_TIMING (`make -C diffco_amd/csrc EXTRA=-DDCX_TIMING OBJ=../../build/obj_t TARGET=../../variants/libdcx_t.so`).
    DCX_LIB=variants/libdcx_t.so python tools/phase_timing.py --batch 1024 [--workload headline]"""
import argparse
import ctypes as Ct
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from diffco_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="headline")
ap.add_argument("--batch", type=int, default=1024)
args = ap.parse_args()
lib = _lib.require_gpu()
dev = torch.device("cuda", 0)
w = bench.make_workload(args.workload, args.batch, dev)
m, q = w["model"], w["q"]
for _ in range(3):
    s, g = m.score_grad_raw(q)
torch.cuda.synchronize()
buf = (Ct.c_ulonglong * 128)()
lib.dcx_debug_read_ts.argtypes = [Ct.POINTER(Ct.c_ulonglong)]
assert lib.dcx_debug_read_ts(buf) == 0
names = ["start", "staged+barrier", "after FK+barrier", "after sweep", "after reduce", "after J^T", "after trig+barrier"]
t0 = buf[0]
print(f"workload {args.workload} B={args.batch} env NW={os.environ.get('DCX_NW')} YS={os.environ.get('DCX_YS')}")
for slot, name in enumerate(names):
    row = [buf[slot * 8 + wv] for wv in range(8)]
    print(f"{name:<18}" + " ".join(f"{(v - t0) if v else -1:>9d}" for v in row))
for j in range(9):
    print(f"joint {j}: chain done {buf[(7 + j) * 8] - t0 if buf[(7 + j) * 8] else -1:>9d}   J^T done "
          f"{buf[(7 + j) * 8 + 1] - t0 if buf[(7 + j) * 8 + 1] else -1:>9d}")
for j, nm in enumerate(["partial row stored", "release fence", "arrival counted", "second fence", "rows re-read", "G staged"]):
    v = buf[(7 + j) * 8 + 2]
    print(f"finish: {nm:<20} {v - t0 if v else -1:>9d}")
print("(cycles of the constant 100 MHz s_memtime/readcyclecounter clock unless the part reports shader clocks)")
