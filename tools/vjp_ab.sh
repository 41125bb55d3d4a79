#!/bin/bash
# tools/vjp_ab.sh — run on the GPU box (via gpurun): the wrench-form FK walks (default build) against the matrix-adjoint
# forms (variants/libdcx_adj.so, built with EXTRA="-DDCX_VJP_MATRIX_ADJOINT -DDCX_FK_DH_MATRIX") on the same box, interleaved:
# bench lines of the latency-bound workloads, the URDF trees, then the phase stamps of one block (variants/libdcx_t.so, -DDCX_TIMING).
set -u
OUT=gpurun_out/r02_vjp_ab.txt
: > $OUT
line() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"  {d['config']['workload'][:12]:<12} step {d['ms_per_step']*1e3:8.2f} us   kernel {d['roofline']['kernel_ms']*1e3:8.2f} us   {d['value']:8.1f} M evals/s   frac {d['roofline']['frac']:.4f}")
PY
}
for round in 1 2; do
  for w in cfg2 cfg2_panda cfg3 cfg3_poly cfg5 headline; do
    for lib in adj wrench; do
      if [ $lib = adj ]; then export DCX_LIB=$PWD/variants/libdcx_adj.so; else unset DCX_LIB; fi
      python bench.py --workload $w --no-cpu-baseline > /tmp/ab_line.json 2>>gpurun_out/r02_vjp_ab.err
      echo -n "round $round $lib" >> $OUT; line /tmp/ab_line.json >> $OUT
    done
  done
done
unset DCX_LIB
echo "# URDF trees (tools/bench_urdf.py), matrix-adjoint then wrench" >> $OUT
DCX_LIB=$PWD/variants/libdcx_adj.so python tools/bench_urdf.py 2>>gpurun_out/r02_vjp_ab.err | sed 's/^/adj    /' >> $OUT
python tools/bench_urdf.py 2>>gpurun_out/r02_vjp_ab.err | sed 's/^/wrench /' >> $OUT
if [ -f variants/libdcx_t.so ]; then
  echo "# phase stamps of one block (wrench forms, -DDCX_TIMING build)" >> $OUT
  DCX_LIB=$PWD/variants/libdcx_t.so python tools/phase_timing.py --workload cfg2 --batch 4096 >> $OUT 2>>gpurun_out/r02_vjp_ab.err
  DCX_LIB=$PWD/variants/libdcx_t.so python tools/phase_timing.py --workload headline --batch 65536 >> $OUT 2>>gpurun_out/r02_vjp_ab.err
fi
cat $OUT
