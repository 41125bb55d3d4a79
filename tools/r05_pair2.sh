#!/bin/bash
# tools/r05_pair2.sh — GPU box: the evidence for pair2 (two rows per packed instruction, D <= 8 direct form) with the full library:
# the whole GPU suite, the driver's bench command, config #4's kernel trace + counters (profiles/pmc_cfg4.json is derived from it).
set -u
R=$PWD; O=$R/gpurun_out/r05i; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -n 3 $O/gpu_tests.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>$O/bench.err
STEPS=12 bash tools/gpu_profile.sh r05i/prof_cfg4 cfg4 > /dev/null 2>&1
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05i/bench_driver_cmd.json").read().strip().splitlines()[-1]); rf=d["roofline"]
print(d["value"], d["ms_per_step"], rf["frac"], {k:(v.get("ms_per_step"), v.get("frac")) for k,v in d["configs"].items()})
PY
tail -n 25 $O/prof_cfg4/summary.txt
