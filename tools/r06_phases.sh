#!/bin/bash
# tools/r06_phases.sh — GPU box: where a small launch's time goes: cycle stamps of every phase of one block
# (libdcx built with -DDCX_TIMING: diffco_amd/libdcx_t.so) for config #2, config #3's 8192 shard and the trajectory kernel (256
# restarts, and the 32-restart shard in its cluster form), next to the same launches' times eager / replayed from a HIP graph
set -u
export DCX_LIB=$PWD/diffco_amd/libdcx_t.so
echo "== config #2 (B = 4096, the 16-configuration tile)"; python tools/phase_timing.py --workload cfg2 --batch 4096 2>&1 | grep -v amdgpu
echo "== config #2 Panda (split launch)"; python tools/phase_timing.py --workload cfg2_panda --batch 4096 2>&1 | grep -v amdgpu | head -12
echo "== config #3's shard (B = 8192, C = 5, split launch)"; python tools/phase_timing.py --workload cfg3 --batch 8192 2>&1 | grep -v amdgpu
echo "== headline (B = 65536)"; python tools/phase_timing.py --workload headline --batch 65536 2>&1 | grep -v amdgpu
echo "== trajectory kernel, 256 restarts"; python tools/traj_phase.py 256 2>&1 | grep -v amdgpu
echo "== trajectory kernel, 32 restarts (cluster form)"; python tools/traj_phase.py 32 2>&1 | grep -v amdgpu
