mkdir -p gpurun_out/r6
( time python -m pytest tests/ -q -m gpu -rs ) > gpurun_out/r6/gpu_tests.txt 2>&1; echo "suite rc=$?"
grep -n "SKIPPED\|^FAILED\|^ERROR" gpurun_out/r6/gpu_tests.txt | cut -c1-200 | head -20
tail -5 gpurun_out/r6/gpu_tests.txt
