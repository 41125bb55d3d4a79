export DCX_LIB=$PWD/diffco_amd/libdcx_dev.so
python -m pytest tests/test_gpu_multiclass_optim.py -x -q -k "not (1-3-6-50 or 1-2-5-64 or 0-8-4-30 or without_a_persistent)" 2>&1 | tail -8
python -m pytest tests/test_gpu_traj.py -q -k "baxter or hinge or single_adam or fused_optimizer or batched or cluster_form_agrees or cluster_rule" 2>&1 | tail -4
for i in 1 2 3; do python bench.py --workload cfg5_c5 --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5_c5 dev', d['ms_per_step'], d['roofline']['kernel_ms'])"; done
python bench.py --workload cfg5 --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 dev', d['ms_per_step'], d['roofline']['kernel_ms'])"
