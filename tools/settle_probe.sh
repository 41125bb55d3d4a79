#!/bin/bash
# tools/settle_probe.sh — developer tool (GPU box): does the length of bench.py's untimed settle phase matter for the FIRST bench process
# after the GPU has idled (the driver's situation)?  Fresh processes, 15 s of idle GPU in front of each, the driver's command.
mkdir -p gpurun_out/r05
for round in 1 2 3; do
  for S in 100 250 600; do
    sleep 15
    DCX_BENCH_SETTLE_MS=$S python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $round settle %4d ms (%5d launches): step %7.2f us  kernel %7.2f us  %7.1f M evals/s  frac %.4f  clock beside the loop %.2f GHz' % (d['settle_ms'], d['settle_steps'], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['value'], d['roofline']['frac'], d['roofline']['clock']['shader_ghz_under_this_load']))"
  done
done
