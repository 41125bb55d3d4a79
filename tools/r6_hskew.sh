pack2() { echo $(( $1 + ($2 << 10) )); }
for rep in 1 2; do for ws in "0 0" "380 340" "400 340" "420 330" "440 330" "400 320" "360 340" "460 320"; do
set -- $ws; export DCX_HESS_SKEW=$(pack2 $1 $2)
echo "== shares $ws"; DCX_HESS_FORM=1 python tools/jac_hess_skew.py 2>&1 | grep -v amdgpu | grep "B=65536\|B=8192" | sed 's/jac [0-9.]* us//;s/DCX_SKEW=None //'
done; done
