for env in "X=1" "DCX_SKEW=0" "DCX_QT=0" "DCX_SKEW=0 DCX_QT=0"; do echo "== $env"; env $env python -m pytest tests/test_gpu_api.py -q -x -k "scipy_constraint_and_drivers" 2>&1 | tail -2; done
