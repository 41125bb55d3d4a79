python -m pytest tests/test_gpu_hess.py -x -q 2>&1 | tail -4
for f in 0 1 -1; do echo "== DCX_HESS_FORM=$f"; DCX_HESS_FORM=$f python tools/jac_hess_skew.py 2>&1 | grep -v amdgpu | sed 's/jac [0-9.]* us//;s/DCX_SKEW=None //'; done
python -m pytest tests -m gpu -q 2>&1 | tail -4
