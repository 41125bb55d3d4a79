#!/bin/bash
# tools/r06_mfma.sh — run on the GPU box (gpurun): the matrix-core A/B re-taken on round 6's kernels, against
# diffco_amd/libdcx_matrix.so (what build() makes with ONLY_WIDTHS="12 16" EXTRA=-DDCX_WITH_MATRIX_FORMS): bench lines with the
# VALU forms and with DCX_MFMA=1 / DCX_XM=1, rocprofv3 matrix-core counters of the MFMA forms (separate --pmc passes), and the
# JSON bench.py reads for `roofline.mfma` (profiles/mfma_contractions.json keeps the round-3 microbenchmarks of the contractions
# in isolation; its `forms` are replaced by what this script measures).
set -u
R=$PWD; O=$R/gpurun_out/r6; mkdir -p $O
export DCX_LIB=$R/diffco_amd/libdcx_matrix.so
T=$O/r06_mfma_ab.txt; : > $T
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s %-46s step %8.2f us  kernel %8.2f us  %8.1f M evals/s  frac %.4f' % ('$1', '$2', d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['value'], d['roofline']['frac']))"; }
echo "== bench.py against libdcx_matrix.so: VALU form vs DCX_MFMA=1 (gradient fold, and K.W for C = 5 / 8, on v_mfma_f32_16x16x4_f32), interleaved ==" >> $T
for w in headline cfg3 "cfg3 --batch 65536" cfg3_poly cfg3_c8; do
  for v in DCX_MFMA=0 DCX_MFMA=1 DCX_MFMA=0 DCX_MFMA=1; do
    env $v timeout 300 python bench.py --workload $w --no-cpu-baseline --no-configs 2>>$O/r06_mfma.err | line $v "$w" >> $T
  done
done
echo >> $T
echo "== expanded form: distance GEMM on the VALU vs on the matrix cores (DCX_XM=1: bf16x3 split operands on v_mfma_f32_16x16x32_bf16) ==" >> $T
for w in headline "headline --batch 1048576 --steps 20" cfg2; do
  for v in DCX_XM=0 DCX_XM=1 DCX_XM=0 DCX_XM=1; do
    env $v timeout 300 python bench.py --workload $w --no-cpu-baseline --no-configs 2>>$O/r06_mfma.err | line $v "$w" >> $T
  done
done
echo >> $T
cd /tmp && export TMPDIR=/tmp
for spec in "headline:DCX_MFMA=1:" "cfg3:DCX_MFMA=1:" "cfg3_c8:DCX_MFMA=1:" "headline:DCX_XM=1:_xm"; do
  w=${spec%%:*}; rest=${spec#*:}; v=${rest%%:*}; tag=${rest#*:}
  env $v timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
      -d $O/r06_pmc_mfma_$w$tag -o bench -- python $R/bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --no-configs > $O/r06_pmc_mfma_$w$tag.log 2>&1
done
cd $R
echo "== rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, per launch ==" >> $T
python tools/mfma_pmc_summary.py $O/r06_pmc_mfma_headline $O/r06_pmc_mfma_cfg3 $O/r06_pmc_mfma_cfg3_c8 $O/r06_pmc_mfma_headline_xm > $O/r06_mfma_pmc.txt 2>&1
cat $O/r06_mfma_pmc.txt >> $T
find $O -name "*.db" -size +2M -delete
cat $T
