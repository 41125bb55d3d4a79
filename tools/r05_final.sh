#!/bin/bash
# tools/r05_final.sh — run on the GPU box (gpurun): the round's closing evidence with the final library.
set -u
R=$PWD; O=$R/gpurun_out/r05f; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; tail -n 4 $O/gpu_tests.txt
: > $O/r05_bench_driver_cmd.jsonl
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 >> $O/r05_bench_driver_cmd.jsonl 2>>$O/bench.err; done
python bench.py > $O/r05_bench_default.json 2>>$O/bench.err
env -u WORLD_SIZE DCX_BENCH_SAME_GPU=1 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/r05_bench_gpus2_selflaunch.json 2>>$O/bench.err
: > $O/r05_bench_forcedist.jsonl
MASTER_PORT=29561 python bench.py --force-dist >> $O/r05_bench_forcedist.jsonl 2>>$O/bench.err
MASTER_PORT=29562 python bench.py --force-dist --scaling strong --workload cfg3 >> $O/r05_bench_forcedist.jsonl 2>>$O/bench.err
bash tools/gpu_profile.sh r05f/prof_headline > /dev/null 2>&1
bash tools/gpu_profile.sh r05f/prof_cfg3_b65536 cfg3 65536 > /dev/null 2>&1
STEPS=12 bash tools/gpu_profile.sh r05f/prof_cfg4 cfg4 > /dev/null 2>&1
python tools/api_latency.py > $O/api_latency_final.txt 2>&1
python - <<'PY'
import json
for f in ("gpurun_out/r05f/r05_bench_driver_cmd.jsonl", "gpurun_out/r05f/r05_bench_default.json", "gpurun_out/r05f/r05_bench_gpus2_selflaunch.json", "gpurun_out/r05f/r05_bench_forcedist.jsonl"):
    for l in open(f):
        if not l.strip(): continue
        d = json.loads(l)
        rf = d["roofline"]
        print(f.split("/")[-1][:28], d["n_gpus"], d["value"], d["ms_per_step"], rf["frac"], rf.get("frac_at_measured_clock"), (rf.get("clock") or {}).get("sclk_mhz_mean"), (d.get("multi") or {}).get("gather"), (d.get("multi") or {}).get("primary"))
        for k, v in (d.get("configs") or {}).items(): print("    ", k, v.get("ms_per_step"), v.get("frac"), v.get("error"))
        if d.get("callers"): print("    callers", json.dumps(d["callers"].get("poly_score_us")))
PY
