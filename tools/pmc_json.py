#!/usr/bin/env python3
"""tools/pmc_json.py <summary.txt> <workload> <batch> <out.json> ["workload description"] — turn the text summary tools/gpu_profile.sh
writes (tools/rocpd_summary.py: kernel-trace stats, FETCH_SIZE / WRITE_SIZE passes with their calibration on known-byte-count
copies, two SQ counter passes) into profiles/pmc_<workload>.json, the file bench.py reads `roofline.traffic` from.  The
dominant kernel is the one with the largest total duration in the trace."""
import json
import re
import sys

path, wl, batch, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
desc = sys.argv[5] if len(sys.argv) > 5 else wl
txt = open(path).read()
sections = {m.group(1).rsplit("/", 1)[-1]: m.group(2) for m in re.finditer(r"^==== (\S+)\n(.*?)(?=^==== |\Z)", txt, re.M | re.S)}
stats = [ln for ln in sections["trace"].split("\n") if ln.startswith('"')][1:]
rows = []
for ln in stats:
    m = re.match(r'"(.*)",(\d+),(\d+),([\d.]+),', ln)
    if m:
        rows.append((int(m.group(3)), m.group(1), int(m.group(2)), float(m.group(4))))
rows.sort(reverse=True)
_, kname, calls, avg_ns = rows[0]
short = re.sub(r"\s+", "", re.sub(r"^void ", "", kname).split("(")[0])


def per_dispatch(section, counter, name_part):
    for ln in sections.get(section, "").split("\n"):
        if ln.startswith(counter + " ") and name_part in ln:
            return float(re.search(r"per_dispatch=(\S+)", ln).group(1))
    return None


key = kname[:60]
fetch, write = per_dispatch("pmc_FETCH_SIZE", "FETCH_SIZE", key), per_dispatch("pmc_WRITE_SIZE", "WRITE_SIZE", key)
cal_f, cal_w = per_dispatch("calib_FETCH_SIZE", "FETCH_SIZE", "calib_copy4"), per_dispatch("calib_WRITE_SIZE", "WRITE_SIZE", "calib_copy4")
GIB_KB = 1048576.0
ff, fw = (cal_f / GIB_KB if cal_f else None), (cal_w / GIB_KB if cal_w else None)
hbm = None
if None not in (fetch, write, ff, fw):
    hbm = int(round(fetch * 1024 / ff + write * 1024 / fw))
ctr = {}
for sec in ("pmc_sq", "pmc_sq2"):
    for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY",
              "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"):
        v = per_dispatch(sec, c, key)
        if v is not None:
            ctr[c] = v
res = {
    "workload": desc, "batch_per_gpu": batch, "kernel": short,
    "kernel_us_avg_under_profiler": round(avg_ns / 1e3, 2), "launches_in_trace": calls,
    "FETCH_SIZE_KB_per_launch_raw": fetch, "WRITE_SIZE_KB_per_launch_raw": write,
    "calibration": f"tools/pmc_calib (1 GiB read + 1 GiB written, same rocprofv3 passes): FETCH_SIZE reports {cal_f:.0f} KB per GiB read = "
                   f"{ff:.4f}x, WRITE_SIZE {cal_w:.0f} KB per GiB = {fw:.4f}x ({path})" if None not in (cal_f, cal_w) else None,
    "hbm_bytes_per_launch": hbm, "counters_per_launch": ctr,
    "clock_note": "the SQ / GRBM cycle counters do not tick at the SIMDs' load-dependent clock (profiles/r05_clock_under_load.txt): "
                  "instruction counts are exact, 'busy' ratios formed from cycle counters are not",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
