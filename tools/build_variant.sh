#!/bin/bash
# Build the GPU library of another git revision as searchlite_amd/lib/libsearchlite_gpu_<tag>.so,
# to time two kernels side by side on ONE box (devices differ by several percent):
#   bash tools/build_variant.sh <git-ref> <tag> [extra compiler flags, e.g. -DSLG_U4_WPB=2];  SLG_LIB_TAG=<tag> python bench.py ...
set -e
REF=$1; TAG=$2; shift 2; EXTRA="$*"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/slg_variant_XXXX)
git -C "$ROOT" archive "$REF" searchlite_amd/csrc include | tar -x -C "$TMP"
cd "$TMP/searchlite_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $EXTRA"
OBJS=""
for kr in 1 2 4 8 16; do
  /opt/rocm/bin/hipcc $FLAGS -DSLG_INST_KREGS=$kr -c slg_score_inst.hip -o k$kr.o &
  OBJS="$OBJS k$kr.o"
done
/opt/rocm/bin/hipcc $FLAGS -c slg_api.hip -o api.o &
PLAN=""
if [ -f slg_plan.cpp ]; then /opt/rocm/bin/hipcc $FLAGS -x c++ -c slg_plan.cpp -o plan.o & PLAN="plan.o"; fi
if [ -f slg_coalesce.hip ]; then /opt/rocm/bin/hipcc $FLAGS -c slg_coalesce.hip -o coalesce.o & PLAN="$PLAN coalesce.o"; fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/searchlite_amd/lib/libsearchlite_gpu_$TAG.so" api.o $PLAN $OBJS
rm -rf "$TMP"
echo "$ROOT/searchlite_amd/lib/libsearchlite_gpu_$TAG.so"
