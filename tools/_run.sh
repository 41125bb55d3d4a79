rm -f gpurun_out/r02_bench_configs.jsonl gpurun_out/r02_bench_forcedist.jsonl
for w in headline cfg2 cfg2_panda cfg3 cfg3_poly cfg4 cfg5; do python bench.py --workload $w >> gpurun_out/r02_bench_configs.jsonl 2>>gpurun_out/r02_bench.err; done
MASTER_PORT=29561 python bench.py --force-dist --scaling strong --workload cfg3 >> gpurun_out/r02_bench_forcedist.jsonl 2>>gpurun_out/r02_bench.err
MASTER_PORT=29562 python bench.py --force-dist >> gpurun_out/r02_bench_forcedist.jsonl 2>>gpurun_out/r02_bench.err
MASTER_PORT=29563 python bench.py --force-dist --workload cfg5 --scaling strong >> gpurun_out/r02_bench_forcedist.jsonl 2>>gpurun_out/r02_bench.err
python - <<'PY'
import json
for f in ("gpurun_out/r02_bench_configs.jsonl","gpurun_out/r02_bench_forcedist.jsonl"):
    for l in open(f):
        d=json.loads(l)
        print(d["config"]["workload"][:10], d["scaling"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], (d.get("cpu_baseline") or {}).get("value"), d.get("multi"))
        for k,v in (d.get("variants") or {}).items(): print("    ",k,v["value"],v["ms_per_step"],v["gather_ms"])
PY
