#!/bin/bash
# tools/r03_mfma.sh [lib] — the matrix-core A/B of round 3 on one GPU box: the corrected co-issue microbenchmark, the three
# contractions VALU vs MFMA (tools/contraction_ubench.hip), the parity tests in the MFMA form and the bench lines of the
# headline / cfg3 / cfg3_poly workloads with and without DCX_MFMA=1.  Binaries are built beforehand into devlibs/.
set -u
OUT=gpurun_out
mkdir -p $OUT
LIB=${1:-$PWD/diffco_amd/libdcx.so}
export DCX_LIB=$LIB
R=$OUT/r03_mfma_ab.txt
: > $R
echo "== tools/mfma_coissue_ubench.hip (fp32 MFMA) ==" >> $R
timeout 300 devlibs/mfma_coissue_f32 >> $R 2>&1
echo >> $R
echo "== tools/mfma_coissue_ubench.hip -DBF16 ==" >> $R
timeout 300 devlibs/mfma_coissue_bf16 >> $R 2>&1
echo >> $R
for occ in 2 4 8; do
  echo "== tools/contraction_ubench.hip, $occ waves per SIMD ==" >> $R
  timeout 300 devlibs/contraction_ubench $occ >> $R 2>&1
  echo >> $R
done
echo "== bench.py, VALU form vs DCX_MFMA=1 ==" >> $R
for w in headline cfg3 cfg3_poly cfg3_c8; do
  for v in "DCX_MFMA=0" "DCX_MFMA=1"; do
    env $v timeout 300 python bench.py --workload $w --no-cpu-baseline --no-configs 2>>$OUT/r03_mfma.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %-12s step %8.2f us   kernel %8.2f us   %8.1f M evals/s   frac %.4f   %s' % ('$v', d['config']['workload'][:12], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['value'], d['roofline']['frac'], d['roofline'].get('kernel','')))" >> $R
  done
done
echo >> $R
echo "== bench.py, expanded form with the distance GEMM on the VALU vs on the matrix cores (DCX_XM=1: bf16x3 split operands) ==" >> $R
for w in "headline" "headline --batch 1048576 --steps 20" "cfg2"; do
  for v in "DCX_XM=0" "DCX_XM=1" "DCX_XM=0" "DCX_XM=1"; do
    env $v timeout 300 python bench.py --workload $w --no-cpu-baseline --no-configs 2>>$OUT/r03_mfma.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s %-40s kernel %9.2f us   %8.1f M evals/s   frac %.4f' % ('$v', '$w', d['roofline']['kernel_ms']*1e3, d['value'], d['roofline']['frac']))" >> $R
  done
done
cat $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "${MFMA_TESTS:-mfma or matrix_cores or matrix_core}" > $OUT/r03_mfma_pytest.txt 2>&1
tail -5 $OUT/r03_mfma_pytest.txt
