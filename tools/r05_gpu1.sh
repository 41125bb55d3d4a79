set -x
mkdir -p gpurun_out/r05
python tools/api_latency.py > gpurun_out/r05/api_latency_before.txt 2>&1
python tools/api_profile.py 50 > gpurun_out/r05/api_profile_before.txt 2>&1
python tools/rq_gap.py > gpurun_out/r05/rq_gap.txt 2>&1
python -m pytest tests/test_gpu_bench_contract.py -x -q > gpurun_out/r05/bench_contract.txt 2>&1
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_bench_contract.py > gpurun_out/r05/gpu_tests.txt 2>&1
tail -5 gpurun_out/r05/bench_contract.txt gpurun_out/r05/gpu_tests.txt
