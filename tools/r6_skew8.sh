export DCX_LIB=$PWD/diffco_amd/libdcx_sk.so
for rep in 1 2; do for sk in 0 550 600 650 700 750; do
export DCX_SKEW8=$sk
for w in "cfg3 --batch 65536" "cfg3_poly --batch 65536" "cfg2_panda" "cfg2_panda --batch 65536" "headline --batch 1048576 --steps 20" "cfg3 --batch 1048576 --steps 20"; do
python bench.py --workload $w --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skew8=$sk', '$w', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done; done
