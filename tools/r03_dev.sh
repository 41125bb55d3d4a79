#!/bin/bash
# tools/r03_dev.sh [tag] — one GPU-box pass of the round-3 development loop with developer libraries
# (devlibs/libdcx_dev.so: widths 12 16 21 24 42, one and five classes): the bitwise walk test, bench lines, phase stamps.
set -u
OUT=gpurun_out
mkdir -p $OUT
TAG=${1:-a}
export DCX_LIB=$PWD/devlibs/libdcx_dev.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_traj.py -x -q -m gpu -k "dh_fk_walks or traj" > $OUT/r03_dev_${TAG}_pytest.txt 2>&1
tail -3 $OUT/r03_dev_${TAG}_pytest.txt
: > $OUT/r03_dev_${TAG}_bench.txt
for w in cfg2 cfg2_panda cfg3 cfg3_poly cfg5 headline; do
  for v in "dev" "dev DCX_JT_WAVES=0"; do
    set -- $v
    lib=$1; shift
    env DCX_LIB=$PWD/devlibs/libdcx_$lib.so "$@" timeout 300 python bench.py --workload $w --no-cpu-baseline --no-configs 2>>$OUT/r03_dev.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s %-12s step %8.2f us   kernel %8.2f us   %8.1f M evals/s   frac %.4f' % ('$v', d['config']['workload'][:12], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['value'], d['roofline']['frac']))" >> $OUT/r03_dev_${TAG}_bench.txt
  done
done
cat $OUT/r03_dev_${TAG}_bench.txt
: > $OUT/r03_dev_${TAG}_phase.txt
export DCX_LIB=$PWD/devlibs/libdcx_t.so
for blk in 0 1; do
  DCX_TS_BLOCK=$blk timeout 120 python tools/phase_timing.py --workload cfg2 --batch 4096 >> $OUT/r03_dev_${TAG}_phase.txt 2>>$OUT/r03_dev.err
done
for blk in 0 1; do
  DCX_TS_BLOCK=$blk timeout 120 python tools/phase_timing.py --workload cfg3 --batch 8192 >> $OUT/r03_dev_${TAG}_phase.txt 2>>$OUT/r03_dev.err
done
timeout 120 python tools/phase_timing.py --workload cfg2_panda --batch 4096 >> $OUT/r03_dev_${TAG}_phase.txt 2>>$OUT/r03_dev.err
timeout 120 python tools/phase_timing.py --workload headline --batch 65536 >> $OUT/r03_dev_${TAG}_phase.txt 2>>$OUT/r03_dev.err
cat $OUT/r03_dev_${TAG}_phase.txt
