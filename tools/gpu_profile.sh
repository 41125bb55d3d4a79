#!/bin/bash
# tools/gpu_profile.sh <tag> [workload [batch]] — run on the GPU box (via gpurun): kernel trace + PMC passes of one of
# bench.py's workloads (default: headline), written under gpurun_out/<tag>/ ; tools/rocpd_summary.py turns them into the text
# summaries committed under profiles/.  PMC passes are separate runs with --kernel-trace only (the pool
# refuses --pmc combined with sys/hip/hsa traces).
set -u
TAG=${1:-prof}
WL=${2:-headline}
BATCH=${3:-0}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --workload $WL --batch $BATCH --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-configs"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $B > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o bench -- $B > $OUT/pmc_$C.log 2>&1
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/calib_$C -o calib -- $R/tools/pmc_calib > $OUT/calib_$C.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/pmc_sq -o bench -- $B > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE GRBM_COUNT \
  --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o bench -- $B > $OUT/pmc_sq2.log 2>&1
cd $R
python tools/rocpd_summary.py $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/calib_FETCH_SIZE $OUT/calib_WRITE_SIZE $OUT/pmc_sq $OUT/pmc_sq2 > $OUT/summary.txt 2>&1
# keep only text (CSV + logs are small; drop any large db)
find $OUT -name "*.db" -size +2M -delete
tail -n 40 $OUT/summary.txt
