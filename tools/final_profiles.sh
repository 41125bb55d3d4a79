#!/bin/bash
# tools/final_profiles.sh — run on the GPU box: the round's closing evidence with the current library: bench lines of every
# workload (tools/_run.sh), rocprofv3 kernel stats + PMC passes of the headline (tools/gpu_profile.sh), kernel stats per config.
set -u
R=$PWD
bash tools/_run.sh > gpurun_out/run_summary.txt 2>&1
bash tools/gpu_profile.sh r02_final > gpurun_out/profile_final.log 2>&1
: > gpurun_out/r02_configs_rocprof_summary.txt
for w in cfg2 cfg3 cfg5; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_cfg_$w -o bench -- python $R/bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r02_cfg_$w.log 2>&1 )
  echo "==== gpurun_out/r02_cfg_$w" >> gpurun_out/r02_configs_rocprof_summary.txt
  f=$(find gpurun_out/r02_cfg_$w -name "*kernel_stats.csv" | head -1)
  echo "# $(basename $f)" >> gpurun_out/r02_configs_rocprof_summary.txt
  head -6 $f | cut -c1-400 >> gpurun_out/r02_configs_rocprof_summary.txt
done
cat gpurun_out/run_summary.txt; tail -40 gpurun_out/r02_final/summary.txt; cat gpurun_out/r02_configs_rocprof_summary.txt
