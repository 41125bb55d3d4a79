for rep in 1 2; do for cfgv in "X=1" "DCX_NW=8" "DCX_NW=8 DCX_SKEW8=650" "DCX_NW=8 DCX_SKEW8=550" "DCX_SKEW=$((500 + (300<<10) + (150<<20)))" "DCX_SKEW=$((470 + (310<<10) + (160<<20)))" "DCX_SKEW=$((490 + (330<<10) + (140<<20)))"; do
for w in "headline" "headline_rq"; do
env $cfgv python bench.py --workload $w --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfgv', '$w', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done; done
