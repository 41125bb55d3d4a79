python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_multiclass_optim.py -q 2>&1 | tail -3
bash tools/r06_final.sh > gpurun_out/r06_final.log 2>&1; tail -120 gpurun_out/r06_final.log
bash tools/r06_phases.sh > gpurun_out/r06_phases.txt 2>&1; wc -l gpurun_out/r06_phases.txt
