for a in "--no-cpu-baseline --steps 10 --warmup 3" "--no-cpu-baseline"; do
python bench.py $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$a', d['value'], d['roofline']['kernel_ms'])
for k,v in d['configs'].items(): print('   ', k, v.get('ms_per_step'), v.get('graph_ms_per_step'), v.get('frac'))
print(d['strong_bound']['cfg3_65536_over_8'], d['strong_bound']['cfg5_256_restarts_over_8'], d['callers']['headline_cold_us']['mean_first'], d['callers']['headline_cold_us']['settled'])
"
done
