export DCX_LIB=$PWD/diffco_amd/libdcx_dev.so
python -m pytest tests/test_gpu_multiclass_optim.py -x -q -k "persistent_launch and not (1-3-6-50 or 1-2-5-64 or 0-8-4-30)" 2>&1 | tail -8
