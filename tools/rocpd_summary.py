#!/usr/bin/env python3
"""Summarise rocprofv3 output for profiles/: kernel-trace stats (from the rocpd sqlite DB or the CSVs)
and PMC counter sums per kernel.  Usage: python tools/rocpd_summary.py <rocprof output dir> [...]"""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def from_db(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x), max(grid_x) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'%':>6} {'vgpr':>5} {'sgpr':>5} "
          f"{'lds':>7} {'wg':>5} {'grid':>9}  kernel")
    for n, k, s, a, mn, mx, vg, sg, lds, wg, grid in rows:
        print(f"{k:>6} {s / 1e3:>12.2f} {a / 1e3:>10.3f} {mn / 1e3:>10.3f} {mx / 1e3:>10.3f} {100 * s / tot:>6.2f} {vg:>5} {sg:>5} "
              f"{lds:>7} {wg:>5} {grid:>9}  {n}")
    try:
        pmc = c.execute("select k.name, p.name, count(*), sum(e.value) from pmc_events e join pmc_info p on e.pmc_id = p.id "
                        "join kernels k on e.event_id = k.id group by k.name, p.name").fetchall()
        for r in pmc:
            print("PMC", r)
    except Exception:
        pass


def from_csv(d):
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        print(f"# {os.path.relpath(f, d)}")
        print(open(f).read().strip())
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        print(f"# {os.path.relpath(f, d)}  (per-kernel sums over dispatches)")
        agg, cnt = defaultdict(float), defaultdict(int)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                key = (row.get("Kernel_Name", "?"), row.get("Counter_Name", "?"))
                agg[key] += float(row.get("Counter_Value", 0) or 0)
                cnt[key] += 1
        for (k, cn), v in sorted(agg.items()):
            print(f"{cn:<28} dispatches={cnt[(k, cn)]:<5} sum={v:<18.6g} per_dispatch={v / cnt[(k, cn)]:<16.6g} {k[:90]}")


for d in sys.argv[1:]:
    print(f"==== {d}")
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    for db in dbs:
        from_db(db)
    from_csv(d)
