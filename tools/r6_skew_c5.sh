pack() { echo $(( $1 + ($2 << 10) + ($3 << 20) )); }
for rep in 1 2; do for ws in "460 300 180" "440 300 190" "420 300 200" "400 300 210" "380 300 220" "420 280 210" "440 280 200" "360 300 230"; do
set -- $ws; export DCX_SKEW=$(pack $1 $2 $3)
for w in "cfg3" "cfg3_poly" "cfg5_c5"; do
python bench.py --workload $w --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w=($ws)', '$w', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done; done
