#!/bin/bash
# tools/r05_final2.sh — second half of the round's closing evidence (GPU box): the whole GPU suite without -x, the N > 1 code
# path on one rank, the matrix-forms build through the parity suite, the split-launch soak.
set -u
R=$PWD; O=$R/gpurun_out/r05g; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -n 3 $O/gpu_tests.txt
: > $O/r05_bench_forcedist.jsonl
MASTER_PORT=29561 python bench.py --force-dist >> $O/r05_bench_forcedist.jsonl 2>>$O/bench.err
MASTER_PORT=29562 python bench.py --force-dist --scaling strong --workload cfg3 >> $O/r05_bench_forcedist.jsonl 2>>$O/bench.err
MASTER_PORT=29563 python bench.py --force-dist --scaling strong --workload cfg5 >> $O/r05_bench_forcedist.jsonl 2>>$O/bench.err
env -u WORLD_SIZE DCX_BENCH_SAME_GPU=1 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/r05_bench_gpus2_selflaunch.json 2>>$O/bench.err
DCX_LIB=$R/devlibs/libdcx_matrix.so python -m pytest tests/test_gpu_parity.py -q > $O/matrix_forms_tests.txt 2>&1; tail -n 3 $O/matrix_forms_tests.txt
python tools/soak_split.py > $O/soak.txt 2>&1; tail -n 3 $O/soak.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_driver_cmd_last.json 2>>$O/bench.err
python - <<'PY'
import json
for f in ("gpurun_out/r05g/r05_bench_forcedist.jsonl", "gpurun_out/r05g/r05_bench_gpus2_selflaunch.json", "gpurun_out/r05g/r05_bench_driver_cmd_last.json"):
    for l in open(f):
        if not l.strip(): continue
        d = json.loads(l); rf = d["roofline"]
        print(f.split("/")[-1][:30], d["n_gpus"], d["value"], d["ms_per_step"], rf["frac"], rf.get("frac_at_measured_clock"), (d.get("multi") or {}).get("gather"), (d.get("multi") or {}).get("primary"))
PY
