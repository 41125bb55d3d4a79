#!/usr/bin/env python3
"""tools/mfma_pmc_summary.py <rocprofv3 output dir>... — per-launch matrix-core counters of the fused score kernel from
`rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
--kernel-trace --output-format csv` passes (tools/r03_final.sh).  Prints one block per directory and a JSON object."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = {}
for d in sys.argv[1:]:
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(f"{d}: no counter_collection.csv")
        continue
    per = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"]
        if "score_kernel" not in k:
            continue
        per[k][row["Counter_Name"]] += float(row["Counter_Value"])
        launches[k].add(row["Dispatch_Id"])
    # (one workload per directory: the kernel that issued matrix instructions, not the score-only / settle variants beside it)
    per = dict(sorted(per.items(), key=lambda kc: kc[1].get("SQ_INSTS_MFMA", 0.0))[-1:])
    for k, c in per.items():
        n = max(len(launches[k]), 1)
        v = {name: val / n for name, val in c.items()}
        mops = v.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) + v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
        busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        gui = v.get("GRBM_GUI_ACTIVE", 0.0)
        # SQ_VALU_MFMA_BUSY_CYCLES sums the matrix pipes' busy cycles over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE sums the
        # launch's length over the 8 XCDs (profiles/README.md, "Reading the SQ counters")
        frac = busy / (gui / 8 * 1024) if gui else None
        rec = {"kernel": k[:110], "launches": n, "SQ_INSTS_MFMA": v.get("SQ_INSTS_MFMA"), "MFMA_MOPS_F32": v.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0), "MFMA_MOPS_BF16": v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0),
               "mfma_flops_per_launch": mops * 512, "MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": gui, "busy_frac": frac}
        out[os.path.basename(d.rstrip("/"))] = rec
        print(os.path.basename(d), json.dumps(rec))
print(json.dumps(out))
