for f in 0 1; do
echo "== DCX_HESS_FORM=$f"; DCX_HESS_FORM=$f python tools/jac_hess_skew.py 2>&1 | grep -v amdgpu | sed 's/jac [0-9.]* us//'
done
DCX_HESS_FORM=1 python -m pytest tests/test_gpu_hess.py -x -q 2>&1 | tail -15
python -m pytest tests/test_gpu_hess.py tests/test_gpu_api.py tests/test_gpu_multiclass_optim.py -x -q 2>&1 | tail -5
