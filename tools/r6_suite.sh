# the whole GPU suite + the default bench line (round 6)
mkdir -p gpurun_out/r6
( time python -m pytest tests/ -q -m gpu -x ) > gpurun_out/r6/gpu_tests.txt 2>&1; echo "suite rc=$?"
tail -6 gpurun_out/r6/gpu_tests.txt
python bench.py > gpurun_out/r6/bench_default.json 2> gpurun_out/r6/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"])
for k,v in d.get("configs",{}).items(): print("  ", k, v.get("ms_per_step"), v.get("frac"), v.get("error"))
print(d.get("strong_bound")); print(d["callers"].get("headline_cold_us"))
PY
