export DCX_LIB=$PWD/diffco_amd/libdcx_sk.so
pack() { echo $(( $1 + ($2 << 10) + ($3 << 20) )); }
for rep in 1 2; do for ws in "250 250 250" "500 300 150" "480 320 150" "520 280 150" "460 300 180" "540 300 120" "500 260 180"; do
set -- $ws; export DCX_SKEW=$(pack $1 $2 $3)
for w in "headline" "cfg5" "cfg5 --batch 1600" "cfg3" "headline --batch 8192" "headline --batch 16384" "headline_rq" "cfg3_poly --batch 65536"; do
python bench.py --workload $w --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w=($ws)', '$w', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done; done
