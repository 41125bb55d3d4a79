#!/bin/bash
# tools/build_variant.sh <name> "<extra compiler flags>" [widths]  — developer builds for A/B runs: devlibs/libdcx_<name>.so with only
# the named feature widths compiled in full (default 12; the others are stubs) and -DDCX_DEV_FAST (one and five classes).
set -e
NAME=$1; EXTRA=${2:-}; WIDTHS=${3:-12}
cd "$(dirname "$0")/../diffco_amd/csrc"
mkdir -p ../../devlibs ../../build/obj_$NAME
make -j${JOBS:-8} OBJ=../../build/obj_$NAME TARGET=../../devlibs/libdcx_$NAME.so ONLY_WIDTHS="$WIDTHS" EXTRA="-DDCX_DEV_FAST $EXTRA" 2>&1 | grep -E "error|Error" || true
ls -la ../../devlibs/libdcx_$NAME.so
