set -u
R=$PWD; O=$R/gpurun_out/r02_mfma; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4 > $O/pytest.txt
timeout 120 tools/mfma_coissue_ubench > $O/coissue.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|VALU_MFMA" | head -40 > $O/counters.txt
B="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline"
for M in 0 1; do
  DCX_MFMA=$M timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_m$M -o bench -- $B > $O/trace_m$M.log 2>&1
  DCX_MFMA=$M timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_m$M -o bench -- $B > $O/pmc_m$M.log 2>&1
done
cd $R
find $O -name "*.db" -size +2M -delete
cat $O/pytest.txt; cat $O/coissue.txt; cat $O/counters.txt | head -20; ls $O/pmc_m1 $O/pmc_m1/* | head; tail -3 $O/pmc_m1.log
