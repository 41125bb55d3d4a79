set -x
mkdir -p gpurun_out/r05
python tools/ab_libs.py v0,v1,v6 cfg3:65536,cfg3,cfg3_poly:65536,headline 2 > gpurun_out/r05/ab_v0_v1_v6.txt 2>&1
python tools/api_latency.py > gpurun_out/r05/api_latency_after1.txt 2>&1
python tools/api_profile.py 50 > gpurun_out/r05/api_profile_after1.txt 2>&1
python -m pytest tests/test_gpu_api.py tests/test_gpu_traj.py -x -q > gpurun_out/r05/tests_api.txt 2>&1
tail -n 5 gpurun_out/r05/tests_api.txt
