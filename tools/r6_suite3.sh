mkdir -p gpurun_out/r6
( time python -m pytest tests/ -q -m gpu -rs ) > gpurun_out/r6/gpu_tests.txt 2>&1; echo "suite rc=$?"
grep -n "SKIPPED" gpurun_out/r6/gpu_tests.txt | cut -c1-220 | sort | uniq -c | head -30
tail -8 gpurun_out/r6/gpu_tests.txt
bash tools/r06_mfma.sh > gpurun_out/r6/mfma_run.log 2>&1; tail -60 gpurun_out/r6/r06_mfma_ab.txt
