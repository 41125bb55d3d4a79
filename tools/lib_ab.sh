#!/bin/bash
# tools/lib_ab.sh <variant.so> <label> [workloads...] — run on the GPU box: bench lines of the default libdcx.so against a
# variant build (DCX_LIB), interleaved, two rounds, on the same box.  Writes gpurun_out/r02_ab_<label>.txt
set -u
VAR=$1; LABEL=$2; shift 2
WL=${@:-cfg2 cfg2_panda cfg3 cfg3_poly cfg5 headline}
OUT=gpurun_out/r02_ab_$LABEL.txt
: > $OUT
for round in 1 2; do
  for w in $WL; do
    for lib in default $LABEL; do
      if [ $lib = default ]; then unset DCX_LIB; else export DCX_LIB=$PWD/$VAR; fi
      python bench.py --workload $w --no-cpu-baseline 2>>gpurun_out/r02_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $round %-8s %-12s step %8.2f us   kernel %8.2f us   %8.1f M evals/s   frac %.4f' % ('$lib', d['config']['workload'][:12], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['value'], d['roofline']['frac']))" >> $OUT
    done
  done
done
unset DCX_LIB
cat $OUT
