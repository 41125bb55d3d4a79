mkdir -p gpurun_out/r6
( time python -m pytest tests/ -q -m gpu -rs ) > gpurun_out/r6/gpu_tests.txt 2>&1; echo "suite rc=$?"
grep -n "SKIPPED\|^FAILED\|^ERROR" gpurun_out/r6/gpu_tests.txt | cut -c1-200 | head -20
tail -4 gpurun_out/r6/gpu_tests.txt
bash tools/r06_final.sh > gpurun_out/r06_final.log 2>&1; grep -v "^    \|^SQ_\|^GRBM" gpurun_out/r06_final.log | head -30 | cut -c1-220
bash tools/r06_phases.sh > gpurun_out/r06_phases.txt 2>&1; wc -l gpurun_out/r06_phases.txt
python tools/two_stream_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r06_two_streams.txt; cat gpurun_out/r06_two_streams.txt
