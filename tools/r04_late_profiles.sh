#!/bin/bash
# tools/r04_late_profiles.sh — rocprofv3 kernel statistics of the two kernels added late in round 4 (run on the GPU box):
# the 16-configuration tile at config #2 (tools/gpu_profile.sh) and dcx_solve beside hipSOLVER's kernels for the same systems.
R=$PWD
bash tools/gpu_profile.sh r04_cfg2_qt cfg2 4096 > /dev/null 2>&1
cp gpurun_out/r04_cfg2_qt/summary.txt gpurun_out/r04_cfg2_qt_rocprof_summary.txt
mkdir -p gpurun_out/r04_solve_prof
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_solve_prof -o solve -- python $R/tools/solve_latency.py 438 2000 > $R/gpurun_out/r04_solve_prof/log.txt 2>&1
cd $R
F=$(find gpurun_out/r04_solve_prof -name "*kernel_stats.csv" | head -1)
(echo "# rocprofv3 --kernel-trace --stats -- python tools/solve_latency.py 438 2000   (dcx_solve: ONE kernel per solve; the library route: the rest)"; grep -v amdgpu gpurun_out/r04_solve_prof/log.txt | tail -3; echo; head -25 "$F") > gpurun_out/r04_solve_rocprof_summary.txt
find gpurun_out/r04_solve_prof -name "*.db" -size +2M -delete
cat gpurun_out/r04_solve_rocprof_summary.txt | cut -c1-200
tail -n 25 gpurun_out/r04_cfg2_qt_rocprof_summary.txt
