export DCX_LIB=$PWD/diffco_amd/libdcx_dev.so
mkdir -p gpurun_out/r6a
python -m pytest tests/test_gpu_multiclass_optim.py -x -q -k "not (1-3-6-50 or 1-2-5-64 or 0-8-4-30 or without_a_persistent)" > gpurun_out/r6a/mc.log 2>&1; echo "mc rc=$?" 
python -m pytest tests/test_gpu_traj.py -q -k "baxter or hinge or single_adam or fused_optimizer or batched or cluster_form_agrees or cluster_rule" > gpurun_out/r6a/traj.log 2>&1; echo "traj rc=$?"
tail -15 gpurun_out/r6a/mc.log; tail -8 gpurun_out/r6a/traj.log
