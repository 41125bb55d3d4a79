mkdir -p gpurun_out/r6
for i in 1 2 3; do python bench.py --workload cfg5_c5 --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5_c5 steps200', d['ms_per_step'], d['roofline']['kernel_ms'])"; done
for i in 1 2; do python bench.py --workload cfg5_c5 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5_c5 steps10', d['ms_per_step'], d['roofline']['kernel_ms'])"; done
python bench.py --workload cfg5 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 steps10', d['ms_per_step'], d['roofline']['kernel_ms'])"
( time python -m pytest tests/ -q -m gpu --deselect tests/test_gpu_bench_contract.py::test_headline_line_carries_every_baseline_config ) > gpurun_out/r6/gpu_tests.txt 2>&1; echo "suite rc=$?"
tail -12 gpurun_out/r6/gpu_tests.txt
