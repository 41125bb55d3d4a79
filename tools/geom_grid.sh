#!/bin/bash
# tools/geom_grid.sh — run on the GPU box: blocks per tile (DCX_YS) x waves per block (DCX_NW) for the latency-bound bench
# workloads; the data behind pick_geometry's split rule, re-taken after the FK walks got shorter.
set -u
OUT=gpurun_out/${GEOM_OUT:-r03_geom_grid.txt}
: > $OUT
for w in ${GEOM_WORKLOADS:-cfg2 cfg3 cfg2_panda}; do
  echo "# $w: step us (kernel us) per DCX_YS x DCX_NW; 'rule' = pick_geometry's own choice" >> $OUT
  python bench.py --workload $w --no-cpu-baseline --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rule        %7.2f (%7.2f)' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))" >> $OUT
  for ys in 1 2 3 4 6 8; do
    for nw in 4 8 16; do
      DCX_YS=$ys DCX_NW=$nw DCX_MIN_ROWS=1 python bench.py --workload $w --no-cpu-baseline --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ys=$ys nw=$nw %7.2f (%7.2f)' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))" >> $OUT
    done
  done
done
cat $OUT
