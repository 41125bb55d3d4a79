for rep in 1 2; do for ab in 0 4 6; do
export DCX_LIB=$PWD/diffco_amd/libdcx_ab$ab.so
for w in "headline" "headline --batch 1048576 --steps 20" "headline --batch 16384" "cfg2"; do
python bench.py --workload $w --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate=$ab', '$w', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done; done
