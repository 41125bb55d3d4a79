pack() { echo $(( $1 + ($2 << 10) + ($3 << 20) )); }
run() { echo "== shares $1 $2 $3 / skew8 $4"; DCX_SKEW=$(pack $1 $2 $3) DCX_SKEW8=$4 python tools/score_only_skew.py 2>&1 | grep -v amdgpu | sed 's/DCX_SKEW=[0-9]* DCX_SKEW8=[0-9]* //'; }
echo "== equal"; DCX_SKEW=0 DCX_SKEW8=0 python tools/score_only_skew.py 2>&1 | grep -v amdgpu | sed 's/DCX_SKEW=[0-9]* DCX_SKEW8=[0-9]* //'
run 350 300 220 540
run 300 280 230 520
run 330 290 220 530
run 320 300 240 550
run 350 270 220 510
run 380 300 200 570
run 360 310 230 540
run 340 300 240 545
run 370 290 210 535
