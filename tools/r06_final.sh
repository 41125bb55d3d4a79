#!/bin/bash
# tools/r06_final.sh — run on the GPU box (gpurun): the round's closing evidence with the final library.
set -u
R=$PWD; O=$R/gpurun_out/r06f; mkdir -p $O
: > $O/r06_bench_driver_cmd.jsonl
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 >> $O/r06_bench_driver_cmd.jsonl 2>>$O/bench.err; done
python bench.py > $O/r06_bench_default.json 2>>$O/bench.err
env -u WORLD_SIZE DCX_BENCH_SAME_GPU=1 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/r06_bench_gpus2_selflaunch.json 2>>$O/bench.err
: > $O/r06_bench_forcedist.jsonl
MASTER_PORT=29561 python bench.py --force-dist >> $O/r06_bench_forcedist.jsonl 2>>$O/bench.err
MASTER_PORT=29562 python bench.py --force-dist --scaling strong --workload cfg3 >> $O/r06_bench_forcedist.jsonl 2>>$O/bench.err
MASTER_PORT=29563 python bench.py --force-dist --scaling strong --workload cfg5 >> $O/r06_bench_forcedist.jsonl 2>>$O/bench.err
bash tools/gpu_profile.sh r06f/prof_headline > /dev/null 2>&1
bash tools/gpu_profile.sh r06f/prof_cfg3_b65536 cfg3 65536 > /dev/null 2>&1
STEPS=12 bash tools/gpu_profile.sh r06f/prof_cfg4 cfg4 > /dev/null 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5_c5 -o bench -- python $R/bench.py --workload cfg5_c5 --steps 192 --warmup 10 --no-cpu-baseline --no-configs > $O/prof_cfg5_c5.log 2>&1 )
for w in cfg2 cfg3 cfg5; do ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o bench -- python $R/bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-configs > $O/prof_$w.log 2>&1 ); done
find $O -name "*.db" -size +2M -delete
python tools/traj_spec_probe.py 2>&1 | grep -v amdgpu > $O/traj_spec_probe.txt
python - <<'PY'
import json
for f in ("gpurun_out/r06f/r06_bench_driver_cmd.jsonl", "gpurun_out/r06f/r06_bench_default.json", "gpurun_out/r06f/r06_bench_gpus2_selflaunch.json", "gpurun_out/r06f/r06_bench_forcedist.jsonl"):
    for l in open(f):
        if not l.strip(): continue
        d = json.loads(l)
        rf = d["roofline"]
        print(f.split("/")[-1][:28], d["n_gpus"], d["value"], d["ms_per_step"], rf["frac"], rf.get("frac_at_measured_clock"), (d.get("multi") or {}).get("gather"), (d.get("multi") or {}).get("primary"))
        for k, v in (d.get("configs") or {}).items(): print("    ", k, v.get("ms_per_step"), v.get("graph_ms_per_step"), v.get("frac"), v.get("error"))
        if d.get("callers"): print("    cold", json.dumps(d["callers"].get("headline_cold_us")), "strong_bound", json.dumps(d.get("strong_bound")))
PY
for d in prof_headline prof_cfg3_b65536 prof_cfg4; do echo "== $d"; tail -n 25 $O/$d/summary.txt; done
for w in cfg5_c5 cfg2 cfg3 cfg5; do echo "== $w"; f=$(ls $O/prof_$w/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -4 "$f"; done
