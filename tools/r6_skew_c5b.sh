pack() { echo $(( $1 + ($2 << 10) + ($3 << 20) )); }
for ws in "460 300 180" "380 300 220" "340 300 240" "320 300 240" "300 290 250" "300 300 260" "280 280 260" "250 250 250"; do
set -- $ws; export DCX_SKEW=$(pack $1 $2 $3)
echo "== shares $ws"; python tools/traj_spec_probe.py 2>&1 | grep -v amdgpu | grep "cfg5_c5"
done
