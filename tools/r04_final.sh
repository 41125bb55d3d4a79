#!/bin/bash
# tools/r04_final.sh — run on the GPU box: round 4's closing evidence with the shipped library.  Everything lands under
# gpurun_out/; the text summaries that are judged get copied to profiles/ afterwards.
set -u
R=$PWD
O=$R/gpurun_out
mkdir -p $O
# 1. the GPU test-suite
timeout 2700 python -m pytest tests -q -m gpu > $O/r04_pytest_gpu.txt 2>&1
tail -4 $O/r04_pytest_gpu.txt
# 2. the driver's command, and the default command
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_driver_cmd.json 2>$O/r04_bench_driver_cmd.err
timeout 600 python bench.py > $O/r04_bench_default.json 2>$O/r04_bench_default.err
python - <<'PY'
import json
for f in ("gpurun_out/r04_bench_driver_cmd.json", "gpurun_out/r04_bench_default.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["settle_steps"], d["cpu_baseline"], d.get("cpu_baseline_torch"))
    for k, v in (d.get("configs") or {}).items():
        print("   ", k, v.get("ms_per_step"), v.get("value"), v.get("frac"), v.get("error"))
PY
# 3. one rank through the N > 1 path (RCCL on one GPU): per-call gather primary, the variants (graph last)
: > $O/r04_bench_forcedist.jsonl
MASTER_PORT=29561 timeout 600 python bench.py --force-dist --no-configs --no-cpu-baseline >> $O/r04_bench_forcedist.jsonl 2>>$O/r04_bench.err
MASTER_PORT=29562 timeout 600 python bench.py --force-dist --no-configs --no-cpu-baseline --scaling strong --workload cfg3 >> $O/r04_bench_forcedist.jsonl 2>>$O/r04_bench.err
MASTER_PORT=29563 timeout 600 python bench.py --force-dist --no-configs --no-cpu-baseline --scaling strong --workload cfg5 >> $O/r04_bench_forcedist.jsonl 2>>$O/r04_bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r04_bench_forcedist.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"][:10], d["scaling"], d["value"], d["ms_per_step"], d.get("multi"))
    for k, v in (d.get("variants") or {}).items():
        print("    ", k, v.get("value"), v.get("ms_per_step"), v.get("gather_ms"), v.get("error"))
PY
# 4. kernel trace + PMC passes: headline, config #3 at 65536 and at its shard, config #4
bash tools/gpu_profile.sh r04_headline > $O/r04_profile_headline.log 2>&1
bash tools/gpu_profile.sh r04_cfg3_b65536 cfg3 65536 > $O/r04_profile_cfg3.log 2>&1
STEPS=20 bash tools/gpu_profile.sh r04_cfg4 cfg4 > $O/r04_profile_cfg4.log 2>&1
for t in r04_headline r04_cfg3_b65536 r04_cfg4; do echo "== $t"; grep -h "score_kernel" $O/$t/summary.txt | cut -c1-230 | head -14; done
# 5. kernel stats per configuration
: > $O/r04_configs_rocprof_summary.txt
for w in cfg2 cfg3 cfg5 "cfg5 --batch 1600" headline_rq; do
  tag=$(echo $w | tr -d ' -')
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_cfg_$tag -o bench -- python $R/bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-configs > $O/r04_cfg_$tag.log 2>&1 )
  echo "==== bench.py --workload $w --steps 100 --warmup 10 (rocprofv3 --kernel-trace --stats)" >> $O/r04_configs_rocprof_summary.txt
  f=$(find $O/r04_cfg_$tag -name "*kernel_stats.csv" | head -1)
  head -6 $f | cut -c1-400 >> $O/r04_configs_rocprof_summary.txt
done
cat $O/r04_configs_rocprof_summary.txt | cut -c1-200
# 6. the trajectory kernel's forms, the model (re)build, the sweep forms
timeout 300 python tools/traj_probe.py 192 256,128,64,32,8 -1 > $O/r04_traj_probe.txt 2>&1
grep -v amdgpu.ids $O/r04_traj_probe.txt
timeout 300 python tools/model_latency.py > $O/r04_model_latency.txt 2>&1
grep -v amdgpu.ids $O/r04_model_latency.txt
timeout 600 python tools/xf_probe.py > $O/r04_xf_probe.txt 2>&1
grep -v amdgpu.ids $O/r04_xf_probe.txt
find $O -name "*.db" -size +2M -delete
