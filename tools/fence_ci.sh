#!/bin/bash
# tools/fence_ci.sh — the hand-overs with their formal fences: builds a developer library with
# -DDCX_HANDOVER_FENCE (release fence in front of the arrival counter instead of the drained write-through stores) and, on
# the GPU box, runs the split-launch parity tests and the soak with it AND with the counter protocol forced
# (DCX_OWNER_POLL=0), so both forms of the counter hand-over stay exercised beside the owner-polls default.
#   here:        bash tools/fence_ci.sh build
#   GPU box:     bash tools/fence_ci.sh run
set -u
case "${1:-run}" in
build)
  make -C diffco_amd/csrc -j8 OBJ=../../build/obj_fence TARGET=../../devlibs/libdcx_fence.so ONLY_WIDTHS="12 21" \
       EXTRA="-DDCX_DEV_FAST -DDCX_HANDOVER_FENCE" > /tmp/fence_build.log 2>&1 && ls -la devlibs/libdcx_fence.so ;;
run)
  export DCX_LIB=$PWD/devlibs/libdcx_fence.so
  for op in 0 -1; do
    echo "== DCX_HANDOVER_FENCE library, DCX_OWNER_POLL=$op"
    DCX_OWNER_POLL=$op timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hess.py -q -m gpu \
        -k "(split or slicing or ragged or graph_capture or owner_polls or hess) and (baxter or panda or cfg2 or cfg3) and not dual and not misc" 2>&1 | tail -2
    DCX_OWNER_POLL=$op timeout 300 python tools/soak_split.py 2>&1 | tail -1
  done ;;
esac
