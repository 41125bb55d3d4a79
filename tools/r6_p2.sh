export DCX_LIB=$PWD/diffco_amd/libdcx_dev.so
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_parity.py -q -x -k "two_rows or config4 or cfg4 or se3 or se2 or planar or cfg1 or ragged or support_slicing" 2>&1 | tail -8
python -m pytest tests/test_gpu_traj.py tests/test_gpu_escape.py -q -k "planar3 or se3 or se2" 2>&1 | tail -5
for i in 1 2 3; do python bench.py --workload cfg4 --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 dev', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
unset DCX_LIB
for i in 1 2 3; do python bench.py --workload cfg4 --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 shipped', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
