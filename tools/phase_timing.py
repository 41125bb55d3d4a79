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
buf = (Ct.c_ulonglong * max(512, lib.dcx_debug_ts_words()))()   # (the library copies ALL its stamp words: phase slots + per-block stamps)
lib.dcx_debug_read_ts.argtypes = [Ct.POINTER(Ct.c_ulonglong)]
assert lib.dcx_debug_read_ts(buf) == 0
NWV = 16
names = {0: "start", 1: "staged+barrier", 6: "after trig+barrier", 2: "after FK+barrier", 3: "after sweep",
         16: "partials in LDS+barrier", 17: "after fold loop", 4: "after reduce", 5: "after J^T"}
t0 = buf[0]
print(f"workload {args.workload} B={args.batch} env NW={os.environ.get('DCX_NW')} YS={os.environ.get('DCX_YS')} FKK={os.environ.get('DCX_FKK')}")
for slot in (0, 1, 6, 2, 3, 16, 17, 4, 5):
    row = [buf[slot * NWV + wv] for wv in range(NWV)]
    print(f"{names[slot]:<24}" + " ".join(f"{(v - t0) if v else -1:>6d}" for v in row))
for j in range(9):
    a, b = buf[(7 + j) * NWV], buf[(7 + j) * NWV + 1]
    print(f"step/joint {j}: chain done {a - t0 if a else -1:>9d}   J^T done {b - t0 if b else -1:>9d}")
for j, nm in enumerate(["partial row stored", "stores drained", "arrival counted", "owner: past the counter", "rows re-read", "G staged"]):
    v = buf[(7 + j) * NWV + 2]
    print(f"finish: {nm:<24} {v - t0 if v else -1:>9d}")
print("(shader cycles, s_memtime; block ts_block of the grid, all 16 waves)")
