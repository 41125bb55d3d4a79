#!/usr/bin/env python3
"""tools/rq_gap.py — developer tool (GPU box): VERDICT r4 item 5.  The driver's line had `headline_rq` at 97.1 us beside the
headline's 86.9 us, the builder's box had them level.  `headline_rq` is the LAST entry of bench.py's `configs`, measured after
config #4's 4.5 ms launches at 100 % VALU and the persistent trajectory kernels: is it the kernel, or the state the GPU is
in?  Measures the two workloads with bench.py's own `measure()` in several orders - first thing, after each other, right
after config #4 / config #5, after an idle second - while a thread samples the GPU's clock and power from sysfs."""
import glob
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)


def _first(pattern):
    hits = sorted(glob.glob(pattern))
    return hits[0] if hits else None


SCLK = _first("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")
POWER = _first("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") or _first("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
TEMP = _first("/sys/class/drm/card*/device/hwmon/hwmon*/temp1_input")


def _read(path, scale):
    try:
        with open(path) as f:
            return float(f.read().strip()) / scale
    except Exception:   # noqa: BLE001
        return float("nan")


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], False

    def run(self):
        while not self.stop:
            self.rows.append((time.perf_counter(), _read(SCLK, 1e6) if SCLK else float("nan"),
                              _read(POWER, 1e6) if POWER else float("nan"), _read(TEMP, 1e3) if TEMP else float("nan")))
            time.sleep(0.01)

    def window(self, t0, t1):
        r = [x for x in self.rows if t0 <= x[0] <= t1]
        if not r:
            return "no samples"
        f = [x[1] for x in r]
        p = [x[2] for x in r]
        return f"sclk {min(f):5.0f}-{max(f):5.0f} MHz (mean {sum(f) / len(f):5.0f}), power mean {sum(p) / len(p):5.0f} W, temp {r[-1][3]:4.0f} C, {len(r)} samples"


print(f"sysfs: sclk={SCLK} power={POWER} temp={TEMP}")
ws = {n: bench.make_workload(n, bench.WORKLOADS[n][4], dev) for n in ("headline", "headline_rq", "cfg4")}
w5 = bench.make_workload("cfg5", 256 * 50, dev)
loops = {n: bench.ScoreLoop(w, dev, 1, "none") for n, w in ws.items()}
loops["cfg5"] = bench.TrajLoop(w5, dev, 1, 256, None)
steps = {"headline": 100, "headline_rq": 100, "cfg4": 12, "cfg5": 200}
smp = Sampler()
smp.start()
order = sys.argv[1:] or ["headline_rq", "headline", "headline_rq", "cfg4", "headline_rq", "headline", "cfg4", "headline", "headline_rq",
                         "cfg5", "headline_rq", "idle", "headline_rq", "headline", "idle", "headline", "headline_rq"]
for name in order:
    if name == "idle":
        torch.cuda.synchronize()
        time.sleep(1.0)
        print("  (1 s idle)")
        continue
    t0 = time.perf_counter()
    wall, km, ns = bench.measure(loops[name], steps[name], 10, dev, False)
    t1 = time.perf_counter()
    print(f"{name:<12} kernel {km * 1e3:8.2f} us  step {wall / steps[name] * 1e6:8.2f} us  ({ns} settle launches)   {smp.window(t0, t1)}")
smp.stop = True
