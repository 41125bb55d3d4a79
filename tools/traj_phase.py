#!/usr/bin/env python3
"""tools/traj_phase.py — developer tool: phase stamps of one iteration (the third) of the persistent trajectory kernel,
block 0, all waves (libdcx built with -DDCX_TIMING).   DCX_LIB=devlibs/libdcx_t.so python tools/traj_phase.py"""
import ctypes as Ct
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from diffco_amd import _lib  # noqa: E402

lib = _lib.require_gpu()
dev = torch.device("cuda", 0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256   # 32: the cluster form (8 workgroups per path), stamps of workgroup (0, 0)
w = bench.make_workload("cfg5", R * 50, dev)
tst, topt, bufs = bench.traj_state(w, R, 50, dev)
st = Ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
_lib.check(lib.dcx_traj_adam_run(w["model"]._h, Ct.byref(tst), Ct.byref(topt), 1, 8, st))
torch.cuda.synchronize()
n = lib.dcx_debug_ts_words()
buf = (Ct.c_ulonglong * n)()
lib.dcx_debug_read_ts.argtypes = [Ct.POINTER(Ct.c_ulonglong)]
assert lib.dcx_debug_read_ts(buf) == 0
names = ["iteration start", "after trig+barrier", "after chain+barrier", "after sweep", "partials+barrier", "after fold / exchange",
         "after path terms / R1", "barrier", "after R1b+barrier", "after R2", "barrier", "after Adam"]
t0 = buf[0]
for slot, nm in enumerate(names):
    row = [buf[slot * 16 + wv] for wv in range(16)]
    print(f"{nm:<24}" + " ".join(f"{(v - t0) if v else -1:>6d}" for v in row))
