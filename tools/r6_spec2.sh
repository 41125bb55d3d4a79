echo "== shipped library (two sweeps per iteration, always)"; python tools/traj_spec_probe.py 2>&1 | grep -v amdgpu
echo "== with the speculation"; DCX_LIB=$PWD/diffco_amd/libdcx_dev.so python tools/traj_spec_probe.py 2>&1 | grep -v amdgpu
