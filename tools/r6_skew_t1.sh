pack() { echo $(( $1 + ($2 << 10) + ($3 << 20) )); }
for rep in 1 2; do for ws in "500 300 150" "480 300 170" "460 300 180" "440 320 180" "460 320 160" "480 280 180" "500 260 180" "420 320 200"; do
set -- $ws; export DCX_SKEW=$(pack $1 $2 $3)
for w in "cfg5"; do
python bench.py --workload $w --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w=($ws)', '$w', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done; done
