R=$GRAFT_REPO_ROOT
for B in 50 1024 8192 65536; do
for f in 0 1; do
( cd /tmp && export TMPDIR=/tmp && DCX_HESS_FORM=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hp_${B}_$f -o x -- python $R/tools/hess_one.py headline $B > /tmp/hp.log 2>&1 )
echo "== B=$B form=$f"; f2=$(find /tmp/hp_${B}_$f -name "*kernel_stats.csv" | head -1); head -4 $f2 | cut -c1-200
done; done
cd $R
DCX_HESS_FORM=1 python tools/jac_hess_skew.py 2>&1 | grep -v amdgpu | sed 's/jac [0-9.]* us//'
DCX_HESS_FORM=1 python -m pytest tests/test_gpu_hess.py -x -q 2>&1 | tail -3
