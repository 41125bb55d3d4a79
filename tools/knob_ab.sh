#!/bin/bash
# tools/knob_ab.sh <ENVVAR> <value> [workloads...] — run on the GPU box: bench lines with and without one developer knob
# (same library, interleaved, two rounds).  Writes gpurun_out/r02_ab_<ENVVAR>.txt
set -u
VAR=$1; VAL=$2; shift 2
WL=${@:-cfg2 cfg2_panda cfg3 cfg3_poly headline}
OUT=gpurun_out/r02_ab_$VAR.txt
: > $OUT
for round in 1 2; do
  for w in $WL; do
    for mode in default "$VAR=$VAL"; do
      if [ "$mode" = default ]; then unset $VAR; else export $VAR=$VAL; fi
      python bench.py --workload $w --no-cpu-baseline 2>>gpurun_out/r02_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $round %-10s %-12s step %8.2f us   kernel %8.2f us   %8.1f M evals/s   frac %.4f' % ('$mode', d['config']['workload'][:12], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['value'], d['roofline']['frac']))" >> $OUT
    done
  done
done
unset $VAR
cat $OUT
