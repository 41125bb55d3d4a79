#!/usr/bin/env python3
"""tools/block_timeline.py — developer tool: when every block of one launch started / began sweeping / ended, and on which CU
(libdcx built with -DDCX_TIMING).  Shows the launch ramp, how many blocks a CU ran side by side and the tail.
    DCX_LIB=devlibs/libdcx_t.so python tools/block_timeline.py --workload headline [--batch B]"""
import argparse
import ctypes as Ct
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from diffco_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="headline")
ap.add_argument("--batch", type=int, default=0)
args = ap.parse_args()
lib = _lib.require_gpu()
dev = torch.device("cuda", 0)
w = bench.make_workload(args.workload, args.batch, dev)
m, q = w["model"], w["q"]
for _ in range(5):
    m.score_grad_raw(q)
torch.cuda.synchronize()
n = lib.dcx_debug_ts_words()
buf = (Ct.c_ulonglong * n)()
lib.dcx_debug_read_ts.argtypes = [Ct.POINTER(Ct.c_ulonglong)]
assert lib.dcx_debug_read_ts(buf) == 0
rows = []
for b in range(4096):
    t0, t1, t2, hw = (buf[512 + 4 * b + k] for k in range(4))
    if t0 == 0 or t2 == 0:
        continue
    xcc, hwid = hw >> 32, hw & 0xffffffff
    cu = (hwid >> 8) & 0xf
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    rows.append((b, t0, t1, t2, xcc, se, sh, cu))
if not rows:
    sys.exit("no stamps")
# the tick counters of different XCDs have different bases: every block's stamps are taken relative to the first start ON ITS XCD
xbase = {}
for r in rows:
    xbase[r[4]] = min(xbase.get(r[4], r[1]), r[1])
rows = [(b, t0 - xbase[x], t1 - xbase[x], t2 - xbase[x], x, se, sh, cu) for (b, t0, t1, t2, x, se, sh, cu) in rows]
base = min(r[1] for r in rows)
end = max(r[3] for r in rows)
print(f"{args.workload} B={w['B']}: {len(rows)} blocks stamped, first start -> last end {end - base} ticks")
starts = sorted(r[1] - base for r in rows)
ends = sorted(r[3] - base for r in rows)
pct = lambda a, p: a[min(len(a) - 1, int(p * len(a)))]
print("block start   (ticks after the first): p0 %d  p50 %d  p90 %d  p100 %d" % (starts[0], pct(starts, .5), pct(starts, .9), starts[-1]))
print("block end                            : p0 %d  p10 %d  p50 %d  p90 %d  p100 %d" % (ends[0], pct(ends, .1), pct(ends, .5), pct(ends, .9), ends[-1]))
life = sorted(r[3] - r[1] for r in rows)
print("block lifetime                       : min %d  p50 %d  max %d" % (life[0], pct(life, .5), life[-1]))
pro = sorted(r[2] - r[1] for r in rows)
print("prologue (start -> sweep start)      : min %d  p50 %d  max %d" % (pro[0], pct(pro, .5), pro[-1]))
per_cu = defaultdict(list)
for r in rows:
    per_cu[r[4:]].append(r)
cnt = sorted(len(v) for v in per_cu.values())
print(f"distinct (xcc, se, sh, cu): {len(per_cu)}; blocks per CU: min {cnt[0]} p50 {pct(cnt, .5)} max {cnt[-1]}")
late = [len([x for x in v if x[1] - base > 20000]) for v in per_cu.values()]
print(f"CUs that started a block more than 20k ticks after the launch began: {sum(1 for x in late if x)}")
worst = max(per_cu.items(), key=lambda kv: max(x[3] for x in kv[1]))
print("the CU that finished last:", worst[0], [(x[0], x[1] - base, x[3] - base) for x in sorted(worst[1], key=lambda x: x[1])])

# the launch in four numbers (ticks relative to the first block start of the same XCD)
sw0 = sorted(r[2] for r in rows)
print("first sweep starts                   : p0 %d  p10 %d  p50 %d" % (sw0[0], pct(sw0, .1), pct(sw0, .5)))
second = sorted(r[1] for r in rows if r[1] > 20000)
if second:
    print("blocks of the later rounds start     : p0 %d  p50 %d  p100 %d  (%d blocks)" % (second[0], pct(second, .5), second[-1], len(second)))
last_end_per_cu = sorted(max(x[3] for x in v) for v in per_cu.values())
print("a CU's last block ends               : p0 %d  p50 %d  p100 %d" % (last_end_per_cu[0], pct(last_end_per_cu, .5), last_end_per_cu[-1]))
last_sweep_start_per_cu = sorted(max(x[2] for x in v) for v in per_cu.values())
print("a CU's last block starts sweeping    : p0 %d  p50 %d  p100 %d" % (last_sweep_start_per_cu[0], pct(last_sweep_start_per_cu, .5), last_sweep_start_per_cu[-1]))
busy = [sum(x[3] - x[1] for x in v) for v in per_cu.values()]
print("sum of block lifetimes per CU        : min %d  p50 %d  max %d   (2 resident at a time: / 2 = %d)" % (min(busy), pct(sorted(busy), .5), max(busy), pct(sorted(busy), .5) // 2))
