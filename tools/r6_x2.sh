for lib in x2b x2a x2b x2a; do
export DCX_LIB=$PWD/diffco_amd/libdcx_$lib.so
for w in headline cfg5 cfg2 "headline --batch 1048576 --steps 20" "headline --batch 8192"; do
python bench.py --workload $w --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$w', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
done
export DCX_LIB=$PWD/diffco_amd/libdcx_x2a.so
python -m pytest tests/test_gpu_parity.py -q -x -k "baxter and not dual" 2>&1 | tail -5
python -m pytest tests/test_gpu_traj.py -q -k "baxter or hinge or single_adam or fused_optimizer or batched or cluster_form_agrees or cluster_rule" 2>&1 | tail -4
