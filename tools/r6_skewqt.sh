export DCX_LIB=$PWD/diffco_amd/libdcx_sk.so
pack() { echo $(( $1 + ($2 << 10) + ($3 << 20) )); }
python -m pytest tests/test_gpu_parity.py -q -x -k "tile_of_16 and not raw" 2>&1 | tail -3
for rep in 1 2; do for ws in "250 250 250" "480 320 150" "400 300 200" "350 300 220" "300 280 230" "440 300 170"; do
set -- $ws; export DCX_SKEW=$(pack $1 $2 $3)
[ "$ws" = "250 250 250" ] && export DCX_SKEW=0
for w in "cfg2" "cfg2 --batch 1024" "cfg2 --batch 256"; do
python bench.py --workload $w --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w=($ws)', '$w', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done; done
