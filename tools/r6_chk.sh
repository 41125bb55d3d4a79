python tools/traj_spec_probe.py 2>&1 | grep -v amdgpu
for w in cfg3 cfg3_poly cfg5_c5 cfg5 "cfg3 --batch 65536"; do python bench.py --workload $w --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d['roofline']['kernel_ms'])"; done
python tools/jac_hess_skew.py 2>&1 | grep -v amdgpu | grep cfg3 | sed 's/hess.*//'
python -m pytest tests/test_gpu_multiclass_optim.py tests/test_gpu_traj.py tests/test_gpu_sharded.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4
