for rep in 1 2; do for sk in 0 -1; do
export DCX_SKEW=$sk
[ "$sk" = "-1" ] && unset DCX_SKEW
for w in "cfg5 --batch 1600" "cfg5" "cfg5_c5" "cfg3" "cfg2" "headline"; do
python bench.py --workload $w --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skew=$sk', '$w', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done; done
python -m pytest tests/test_gpu_traj.py tests/test_gpu_multiclass_optim.py -q 2>&1 | tail -3
